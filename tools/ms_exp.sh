set -e
cp sg-slam_amd/lib/libsgs_hip.so /tmp/orig.so
for v in t64_s0 t64_s4 t32_s1; do
  cp build/var/libsgs_$v.so sg-slam_amd/lib/libsgs_hip.so
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams 2 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$v.json')); print('$v', d['value'], d['roofline']['avg_launch_us'], d['multi_stream']['value'])"
done
cp /tmp/orig.so sg-slam_amd/lib/libsgs_hip.so
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams 2 > gpurun_out/bench_t64_s1.json 2> gpurun_out/bench_t64_s1.err
python -c "
import json; d=json.load(open('gpurun_out/bench_t64_s1.json')); print('t64_s1', d['value'], d['roofline']['avg_launch_us'], d['multi_stream']['value'])"
