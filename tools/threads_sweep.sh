for C in config3 config5; do for T in 1024 896 768 640 512; do
  echo "== $C THX_THREADS=$T"
  THX_VERBOSE=1 THX_THREADS=$T python tools/quick_bench.py 1000000 $C 2>&1 | grep -v "^{" | sort | uniq | cut -c1-130
done; done
