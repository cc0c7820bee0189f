#!/bin/bash
# FIRST CONTACT with a multi-GPU node, as one command with known answers.  Nothing in this repository has ever moved a
# byte between two devices (the pool's boxes have one GPU); every step below is gated on committed oracle digests or on
# the program's own known answers, writes its result under profiles/<tag>/ in the SCALE record's shape, and the script
# goes on to the next step whatever a step's verdict — so that one session on a node ranks the three transports and
# yields the scaling curve, or says exactly which step broke.
#
#   usage: bash bench/first_node.sh [tag=node1] [max_gpus=8]
#
#   (a) two plain-C processes on TWO devices: hodor_dist_ntt_natural_dev, hodor_dist_lde_commit_dev (both tree formats) and
#       10 000 generations of hodor_dist_ntt_forward_dev per peer-mapped transport, every generation compared with its
#       payload's known answer — the flag protocol and assumption A1 of DESIGN §6 (acknowledged peer stores across a
#       kernel boundary) between real devices for the first time           -> scale_dist2.txt
#   (b) the library's schedules over the peer-mapped transports at world 2 / 4 / 8, one process per GPU
#       (tests/dist_worker.py: natural-order transforms, split-phase pairs, LDE by cosets + commit, every rank checking
#       its share against the single-device result it computes for itself)                               -> scale_pytest.txt
#   (c) bench.py --gpus N, N = 1, 2, 4, 8, once per transport (RCCL through the library, RCCL through torch, direct
#       stores, copy engines), every line gated on the oracle's digest of the 2^24 N-point transform  -> scale_<transport>.json
#   (d) the north-star sweep at N = max: totals 2^26 .. 2^30 (2^23 .. 2^27 points per rank)            -> scale_sweep.json
#   (e) the prediction committed beforehand (profiles/r06/scale_prediction.json) beside the measured curve -> scale_vs_prediction.txt
TAG=${1:-node1}
MAXG=${2:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NG" | tee "$OUT/scale_devices.txt"
rocm-smi --showtopo >> "$OUT/scale_devices.txt" 2>&1
[ "$NG" -lt 2 ] && { echo "this is not a multi-GPU node"; exit 3; }
[ "$MAXG" -gt "$NG" ] && MAXG=$NG
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1 || { echo "build failed"; exit 9; }

# ---- (a) two C processes, two devices
gcc -std=c11 -O2 tests/host_c/test_dist2.c -Lhodor_amd -l:libhodor_gpu.so -Wl,-rpath,$ROOT/hodor_amd -o /tmp/test_dist2 &&
  ( time timeout 1800 /tmp/test_dist2 10000 0 1 ) > "$OUT/scale_dist2.txt" 2>&1
echo "(a) test_dist2 on devices 0 and 1: $(grep -c '0 mismatches' "$OUT/scale_dist2.txt") of 4 soaks clean, $(grep -c 'all tests passed' "$OUT/scale_dist2.txt") x 'all tests passed'"

# ---- (b) the Python schedules on RCCL, one process per GPU
for N in 2 4 8; do
  [ "$N" -gt "$MAXG" ] && continue
  for T in direct copy; do
    ( time HODOR_DIST_DEVICE_PER_RANK=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
        --master-port $((29600 + N)) tests/dist_worker.py $T 22 18 8 ) >> "$OUT/scale_pytest.txt" 2>&1
    echo "(b) world $N, transport $T: exit $? (tests/dist_worker.py, one device per rank)" | tee -a "$OUT/scale_pytest.txt"
  done
done

# ---- (c) bench.py per transport and N
run_bench() {   # name, N, extra args...
  local name=$1 N=$2; shift 2
  local line
  if [ "$N" = 1 ]; then line=$(timeout 1200 python bench.py --gpus 1 "$@" 2>>"$OUT/scale_${name}.err" | grep '^{' | tail -1)
  else line=$(timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) \
                 bench.py --gpus $N "$@" 2>>"$OUT/scale_${name}.err" | grep '^{' | tail -1); fi
  echo "${line:-{\"n_gpus\": $N, \"error\": \"no line (see scale_${name}.err)\"\}}" >> "$OUT/scale_${name}.jsonl"
}
for T in native torch direct copy; do
  rm -f "$OUT/scale_${T}.jsonl"
  for N in 1 2 4 8; do
    [ "$N" -gt "$MAXG" ] && continue
    if [ "$N" = 1 ]; then run_bench $T 1 --no-cpu-baseline; else run_bench $T $N --exchange $T; fi
  done
  python3 - "$OUT/scale_${T}.jsonl" "$T" > "$OUT/scale_${T}.json" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r["value"] for r in rows if r.get("n_gpus") == 1 and "value" in r), None)
out = {"transport": sys.argv[2], "metric": "ntt_field_elems_per_sec", "scaling": "weak", "runs": []}
for r in rows:
    e = {"n_gpus": r.get("n_gpus"), "value": r.get("value"), "ms_per_step": r.get("ms_per_step"), "error": r.get("error"),
         "checks": r.get("checks"), "exchange": (r.get("config") or {}).get("exchange"), "retreated": r.get("retreated")}
    if base and r.get("value"): e["speedup_vs_1"] = r["value"] / base
    out["runs"].append(e)
print(json.dumps(out, indent=1))
PY
  echo "(c) transport $T: $(python3 -c "import json;d=json.load(open('$OUT/scale_${T}.json'));print(', '.join('N=%s %.2fx' % (r['n_gpus'], r.get('speedup_vs_1', 0)) for r in d['runs']))")"
done

# ---- (d) the sweep at N = MAXG with the library's default transport
rm -f "$OUT/scale_sweep.jsonl"
LP=$(python3 -c "print(($MAXG).bit_length() - 1)")
for TOTAL in 26 27 28 29 30; do
  LOGN=$((TOTAL - LP))
  line=$(timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $MAXG --master-addr 127.0.0.1 --master-port $((29900 + TOTAL)) \
           bench.py --gpus $MAXG --log-n $LOGN --steps 20 --warmup 5 --no-extra 2>>"$OUT/scale_sweep.err" | grep '^{' | tail -1)
  echo "${line:-{\"log_total\": $TOTAL, \"error\": \"no line\"\}}" >> "$OUT/scale_sweep.jsonl"
done
python3 -c "
import json
rows=[json.loads(l) for l in open('$OUT/scale_sweep.jsonl') if l.strip()]
json.dump({'n_gpus': $MAXG, 'rows': [{'log_total': r.get('config',{}).get('log_total', r.get('log_total')), 'value': r.get('value'), 'ms_per_step': r.get('ms_per_step'), 'hbm_frac_per_rank': (r.get('roofline') or {}).get('frac'), 'checks': r.get('checks'), 'error': r.get('error')} for r in rows]}, open('$OUT/scale_sweep.json','w'), indent=1)"

# ---- (e) against the prediction
python3 - "$OUT" "$ROOT/profiles/r06/scale_prediction.json" > "$OUT/scale_vs_prediction.txt" <<'PY'
import json, os, sys
out, pred = sys.argv[1], json.load(open(sys.argv[2]))
print("N   predicted (overlapped .. serial)      measured per transport")
for N in ("2", "4", "8"):
    p = pred["speedup_vs_1"][N]
    cells = []
    for T in ("native", "torch", "direct", "copy"):
        f = os.path.join(out, "scale_%s.json" % T)
        if not os.path.exists(f): continue
        r = next((r for r in json.load(open(f))["runs"] if str(r.get("n_gpus")) == N), None)
        cells.append("%s %s" % (T, ("%.2fx" % r["speedup_vs_1"]) if r and r.get("speedup_vs_1") else "-"))
    print("%s   %.2fx .. %.2fx                        %s" % (N, p["overlapped"], p["serial"], "   ".join(cells)))
PY
cat "$OUT/scale_vs_prediction.txt"
