for C in config3 config5; do for M in 1 16 32 48 64; do
  echo "== $C THX_TAKE_MIN=$M"
  THX_TAKE_MIN=$M python tools/quick_bench.py 1000000 $C | cut -c1-130
done; done
