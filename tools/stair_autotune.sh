set -e
mkdir -p gpurun_out
for b in 357 576 640 704; do echo "== B=$b"; RS_B=$b timeout -k 10 200 python tools/autotune_probe.py f32w bf16x3 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_stair_autotune.txt 2>&1
cat gpurun_out/r06_stair_autotune.txt
