set -e
out=gpurun_out/r03e
mkdir -p $out
for d in 4 8 16; do for w in 6144 8192 12288; do
  echo "== BVQ_ONEPASS_DEPTH=$d BVQ_ONEPASS_UNITS=$w" >> $out/ab.txt
  BVQ_ONEPASS_DEPTH=$d BVQ_ONEPASS_UNITS=$w python tools/onepass_ab.py --rounds 6 2>&1 | grep absmax >> $out/ab.txt
done; done
cat $out/ab.txt
