#!/bin/bash
# average duration of every fused-kernel instantiation in one bench command, per engine variant (rocprofv3 --kernel-trace;
# the last 20 launches of each kernel = the timed, settled steps).  Usage: kernel_times.sh "<bench args>" name...   (GPU box)
R=$GRAFT_REPO_ROOT; BA="$1"; shift
export TMPDIR=/tmp; cd /tmp
for n in "$@"; do
  OUT=/tmp/kt_$n; rm -rf $OUT
  LIB=$R/build/variants/libgel_$n.so; [ "$n" = main ] && LIB=$R/gelato_amd/libgelato_amd.so
  GELATO_AMD_LIB=$LIB timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o k -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $BA > $OUT.json 2> $OUT.err
  python3 - $OUT $n $OUT.json <<'PY'
import csv, glob, json, re, sys
root, name = sys.argv[1], sys.argv[2]
d = {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            m = re.search(r"eval_kernel<([^>]*)>", r["Kernel_Name"])
            d.setdefault(m.group(1), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
try:
    line = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])
    v = "%.4g evals/s, ms_per_step %.4f" % (line["value"], line["ms_per_step"])
except Exception as e:
    v = "no bench line (%s)" % e
print("%-12s %s" % (name, v))
tot = 0.0
for k, v in sorted(d.items()):
    n_all = len(v)
    v = sorted(v)[-20:]
    avg = sum(b - a for a, b in v) / len(v) / 1e6
    tot += avg
    print("   <%s>  %.4f ms  (mean of the last %d of %d launches)" % (k, avg, len(v), n_all))
print("   sum %.4f ms" % tot)
PY
done
