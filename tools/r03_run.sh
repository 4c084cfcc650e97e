out=gpurun_out/r03s
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_cpp_autograd.py tests/test_gpu_two_ranks.py tests/test_gpu_variants.py tests/test_kl_threshold_golden.py tests/test_calibration_known_answers.py -q -m gpu -x > $out/tests.txt 2>&1 || { tail -40 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
python bench.py --steps 100 --warmup 20 --shard-path --no-cpu-baseline --act-shape 32,512,56,56 2>$out/shard32.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shard32', d['ms_per_step'], d['value'], d['config']['rccl_ranks'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 2>$out/tdr.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('under torch.distributed.run, 1 rank:', d['value'], d['ms_per_step'], d['scaling'])"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof32 -o shard32 -- python3 bench.py --steps 50 --warmup 10 --shard-path --no-cpu-baseline --act-shape 32,512,56,56 > $out/prof32.json 2> $out/prof32.err
python - <<PY
import sqlite3, glob, re
db = sqlite3.connect(glob.glob('$out/prof32/*.db')[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
n = 16
for (k, s, e) in rows[-n:]:
    print('%-70s start %+9.1f us  dur %7.1f us' % (re.sub(r'\(.*', '', k)[:70], (s - rows[-n][1]) / 1e3, (e - s) / 1e3))
PY
