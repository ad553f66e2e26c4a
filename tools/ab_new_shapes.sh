# the library of the previous commit (riser_amd/lib/liblibriser_amd_head.so.so, built from a stash) against the tree's, alternating
# processes on one box: the 192- / 320- / 384-row ring shapes, the F8 kernel's NT = 6 deferral
A=riser_amd/lib/liblibriser_amd_head.so.so; B=riser_amd/lib/libriser_amd.so
out=gpurun_out/r06_ab_new_ring_shapes.txt; : > $out
for dt in bf16x3 f16xf8; do
  for cfg in "512 16000 0" "512 16000 1" "357 8615 0" "300 16000 0" "576 16000 0" "704 16000 0"; do
    set -- $cfg
    echo "== $dt B=$1 L=$2 mixed=$3" >> $out
    if [ "$3" = 1 ]; then export RS_MIXED=1; else unset RS_MIXED; fi
    RS_DT=$dt RS_B=$1 RS_L=$2 timeout -k 10 300 python tools/ab_libs.py $A $B 2>&1 | grep median >> $out
  done
done
cat $out
