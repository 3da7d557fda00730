for cfg in "256 4" "128 4" "256 16" "128 16" "200 8" "256 4"; do
  set -- $cfg
  echo "tiles<$1 ktiles>=$2: $(BMT_SPLITK_TILES=$1 BMT_SPLITK_MIN_KTILES=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
