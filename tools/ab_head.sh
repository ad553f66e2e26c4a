# generic A/B of the previous commit's library (riser_amd/lib/liblibriser_amd_head.so.so: `git stash; python -c "from riser_amd import
# build as B; B.build(lib_name='libriser_amd_head.so')"; git stash pop; python -m riser_amd.build`) against the tree's:
#   RS_DT=f32w bash tools/ab_head.sh "512 16000 0" "576 16000 0" ...       (B L mixed)
A=riser_amd/lib/liblibriser_amd_head.so.so; B=riser_amd/lib/libriser_amd.so
for cfg in "$@"; do
  set -- $cfg
  echo "== ${RS_DT:-f32} B=$1 L=$2 mixed=$3"
  if [ "$3" = 1 ]; then export RS_MIXED=1; else unset RS_MIXED; fi
  RS_B=$1 RS_L=$2 timeout -k 10 300 python tools/ab_libs.py $A $B 2>&1 | grep median
done
