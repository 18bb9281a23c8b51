for L in 4 3 2 1; do
  echo "== THX_LEAF_MAX=$L"
  THX_LEAF_MAX=$L python tools/query_bench.py 1048576 | tail -4 | head -2
  THX_LEAF_MAX=$L python tools/quick_bench.py 10000000 config2
done
