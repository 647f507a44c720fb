for z in f64 f32 c128 c64; do
  for m in 7 9 10; do
    for mode in poly direct; do
      line="$z m=$m $mode :"
      for lib in "$@"; do
        t=$(NUFFT_LIB_PATH=$lib NUFFT_INTERP_MARCH=2 python scripts/perf_probe.py --mode $mode --z $z --m $m --reps 3 2>&1 | grep -E "t2_interp" | awk '{print $2}')
        line="$line  ${t:-NA}"
      done
      echo "$line"
    done
  done
done
