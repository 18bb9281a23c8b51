python tools/query_bench.py 1048576 | tail -4
for C in 256 1024 4096; do for T in 32 48 56; do
  echo "== chunk=$C thr=$T"
  THX_QUERY_CHUNK=$C THX_QUERY_THR=$T python tools/query_bench.py 1048576 | tail -4 | head -2
done; done
