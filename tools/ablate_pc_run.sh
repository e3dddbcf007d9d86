for v in "" pc_no_network pc_no_pose pc_neither; do
  if [ -n "$v" ]; then export MPPI_SO_PATH=$PWD/build/variants/libmppi_hip_$v.so; else unset MPPI_SO_PATH; fi
  echo "== ${v:-full}"; python bench.py --workload nnspeed --no-cpu-baseline --no-subrecords --min-time 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('kernel_us'))"
done
