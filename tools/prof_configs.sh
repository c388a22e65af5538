# rocprofv3 kernel stats of the OTHER claimed bench configurations on the current code (the headline: collect_profiles.sh):
#   TAG=r03_d bash tools/prof_configs.sh      (on the GPU box; <TAG>_<config>_kernel_stats.csv + _bench.json land in gpurun_out/)
TAG=${TAG:-r03_d}; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
run() {   # name, bench flags...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${name}_stats -- python $R/bench.py --cpu-sample 0 --sub none --steps 2 "$@" > $O/${TAG}_${name}_bench.json 2> $O/${TAG}_${name}.err || return 1
  cp $(ls $O/${TAG}_${name}_stats/*/*kernel_stats.csv) $O/${TAG}_${name}_kernel_stats.csv && rm -rf $O/${TAG}_${name}_stats
  python -c "import json; d=json.load(open('$O/${TAG}_${name}_bench.json')); print('$name', d['value'])"
}
run bf16 --dtype bf16 && run f32 --dtype f32 --chunk 4096 --steps 1 && run f32split --dtype f32split && run fpg4 --frames-per-group 4 &&
run both --extractor resnet50+inception3 && run config2 --config 2 --steps 1
