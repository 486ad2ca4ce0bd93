# same-box A/B of the whole step with two builds of the library (libssd_hip_base.so / libssd_hip_new.so), interleaved
for rep in 1 2 3; do
for lib in libssd_hip_base.so libssd_hip_new.so; do
  SSD_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['value'])"
done
done
