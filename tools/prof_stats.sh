# rocprofv3 kernel stats of one bench.py run (headline only): TAG=r03a DTYPE=f16x2 bash tools/prof_stats.sh
TAG=${TAG:-r03a}; DTYPE=${DTYPE:-f16x2}; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python $R/bench.py --dtype $DTYPE --cpu-sample 0 --sub none > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd $R; cp $(ls $O/${TAG}_stats/*/*kernel_stats.csv) $O/${TAG}_kernel_stats.csv && rm -rf $O/${TAG}_stats
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/${TAG}_kernel_stats.csv")))
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
print(json.load(open("$O/${TAG}_bench.json"))["value"])
PY
