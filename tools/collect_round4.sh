# Round-4 evidence in one gpurun call (on the GPU box; everything lands in gpurun_out/):
#   bash tools/collect_round4.sh            -> ${TAG}_* (headline: kernel stats, PMC traffic, SQ counters), the other
#   configurations' kernel stats (training leg, both trunks, configs[2], per-frame groups, bf16), the full bench line
TAG=${TAG:-r04_j}; R=/root/repo; O=$R/gpurun_out; mkdir -p $O
TAG=$TAG DTYPE=f16x2 bash $R/tools/collect_profiles.sh > $O/${TAG}_collect.txt 2>&1; tail -3 $O/${TAG}_collect.txt
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench flags...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${name}_stats -- python $R/bench.py --cpu-sample 0 --sub none "$@" > $O/${TAG}_${name}_bench.json 2> $O/${TAG}_${name}.err || return 1
  cp $(ls $O/${TAG}_${name}_stats/*/*kernel_stats.csv) $O/${TAG}_${name}_kernel_stats.csv && rm -rf $O/${TAG}_${name}_stats
  python -c "import json; d=json.load(open('$O/${TAG}_${name}_bench.json')); print('$name', d['value'])"
}
run train --config 4 --steps 10 && run both --steps 2 --extractor resnet50+inception3 && run config2 --config 2 --steps 1 &&
run fpg1 --steps 2 --frames-per-group 1 && run bf16 --steps 2 --dtype bf16
cd $R && python tools/inception_layer_table.py > $O/${TAG}_inception_layer_table.txt 2>&1
python bench.py > $O/${TAG}_bench_full.json 2> $O/${TAG}_bench_full.log; grep "sub-result\|headline\|cross" $O/${TAG}_bench_full.log
