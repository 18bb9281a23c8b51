for M in 1 8 16 32 48 64; do
  echo "== THX_TAKE_MIN=$M"
  THX_TAKE_MIN=$M python tools/quick_bench.py 10000000 config2
done
