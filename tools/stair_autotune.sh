# rs_autotune (every tile shape of the tables timed on each layer's real input) at the batch sizes of the staircase
set -e
mkdir -p gpurun_out
DTS=${RS_DTS:-"f32w bf16x3"}
for b in ${RS_BS:-357 576 640 704}; do echo "== B=$b"; RS_B=$b timeout -k 10 300 python tools/autotune_probe.py $DTS 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_stair_autotune.txt 2>&1
cat gpurun_out/r06_stair_autotune.txt
