# head + tail splits forced on (RS_TAIL_MARGIN) at the batch sizes of the staircase: per-layer times next to the planner's own
mkdir -p gpurun_out
out=gpurun_out/r06_stair_tail_margin.txt; : > $out
for b in 576 640 704; do
  for mg in 0 1.05 1.3; do
    echo "== B=$b f32w RS_TAIL_MARGIN=$mg" >> $out
    RS_B=$b RS_TAIL_MARGIN=$mg RS_TAIL_DEBUG=1 timeout -k 10 120 python tools/layer_times.py f32w 2>&1 | grep -v amdgpu.ids | sort -u >> $out
  done
  for mg in 0 1.05 1.3; do
    echo "== B=$b bf16x3 RS_RING_TAIL_SPLIT=1 RS_TAIL_MARGIN=$mg" >> $out
    RS_B=$b RS_RING_TAIL_SPLIT=1 RS_TAIL_MARGIN=$mg RS_TAIL_DEBUG=1 timeout -k 10 120 python tools/layer_times.py bf16x3 2>&1 | grep -v amdgpu.ids | sort -u >> $out
  done
done
