# rocprofv3 kernel stats of the f16x2 headline with and without the 3-byte storage of the inner block outputs (same box):
#   bash tools/prof_p8.sh      -> gpurun_out/p8{on,off}_kernel_stats.csv
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for M in on off; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p8${M}_stats -- python $R/bench.py --p8 $M --steps 2 --cpu-sample 0 --sub none > $O/p8${M}_bench.json 2> $O/p8${M}_bench.err || exit 1
  cp $(ls $O/p8${M}_stats/*/*kernel_stats.csv) $O/p8${M}_kernel_stats.csv && rm -rf $O/p8${M}_stats
done
cd $R
python - <<PY
import csv, json
for m in ("on", "off"):
    rows = list(csv.DictReader(open("$O/p8%s_kernel_stats.csv" % m)))
    print("==== p8", m, json.loads(open("$O/p8%s_bench.json" % m).read().strip().splitlines()[-1])["value"])
    for r in rows[:22]:
        print(f"{r['Name'][:96]:96s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['MinNs'])/1e3:9.1f} {float(r['MaxNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}%")
PY
