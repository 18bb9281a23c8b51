python tools/query_bench.py 1048576 one | tail -1
for D in 32 24 16; do echo "== depth=$D (u16)"; THX_QUERY_DEPTH=$D python tools/query_bench.py 1048576 one | tail -1; done
