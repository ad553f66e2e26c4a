set -e
mkdir -p gpurun_out
for b in 512 576 640 704 357; do echo "== B=$b"; RS_B=$b timeout -k 10 120 python tools/layer_times.py f32w bf16x3; done > gpurun_out/r06_stair_layers.txt 2>&1
cat gpurun_out/r06_stair_layers.txt
