# Texture-path counters of the contraction kernel (tools/conv_one.py, 4096 frames, 3 launches): small rocprofv3 --pmc
# passes (some counter combinations abort the profiler on this image: every pass has its own short timeout).
# usage: bash tools/ta_counters.sh "4096 14 256 256 3"            (contraction kernel, tools/conv_one.py)
#        bash tools/ta_counters.sh "4096 3136 64 256" conv1x1    (one-pass 1x1 kernel, tools/conv1x1_one.py)
R=/root/repo; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
shape=${1:-"4096 14 256 256 3"}
which=${2:-igemm}
prog=$R/tools/conv_one.py; [ "$which" = conv1x1 ] && prog=$R/tools/conv1x1_one.py
tag=${which}_$(echo $shape | tr ' ' '_')
i=0
for P in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 5 45 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/ta_tmp -- python $prog $shape > /dev/null 2> $O/ta_err.txt || echo "pass $i failed: $P"
  f=$(ls $O/ta_tmp/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $O/ta_${tag}_q$i.csv
  rm -rf $O/ta_tmp
done
cd $R && KFILTER=$which python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/ta_*_q*.csv')):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if os.environ.get('KFILTER', 'igemm') not in r['Kernel_Name']:
            continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(os.path.basename(f), {k: (round(v / n[k]), n[k]) for k, v in agg.items()})
PY
