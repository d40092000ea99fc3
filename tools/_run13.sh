cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
cp svt_speechbrain_amd/libsvt_mi355.so /tmp/new.so
for i in 1 2; do
  cp /tmp/new.so svt_speechbrain_amd/libsvt_mi355.so
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r02m/new$i.json 2>/dev/null
  cp svt_speechbrain_amd/libsvt_old.so svt_speechbrain_amd/libsvt_mi355.so
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r02m/old$i.json 2>/dev/null
done
cp /tmp/new.so svt_speechbrain_amd/libsvt_mi355.so
for f in new1 old1 new2 old2; do python -c "
import json; r=json.load(open('gpurun_out/r02m/$f.json')); print('$f', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['ms_per_step'])"; done
