for i in 1 2 3; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export INNFER_LIB=$PWD/innfer_amd/lib/libinnfer_amd_prev.so; else unset INNFER_LIB; fi
    python3 bench.py --no-extras --sharded-steps 0 --steps 10 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; [print('$lib', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
