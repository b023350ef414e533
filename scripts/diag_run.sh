for v in "" 64 128; do
  if [ -z "$v" ]; then lib=""; else lib="variants/libdiag$v.so"; fi
  echo "=== diag ${v:-0}"
  CFD_AMD_LIB=$lib timeout -k 10 120 python scripts/layer_bench.py --net U --batch 128 --modes fwd --only d4,d5 --bias 1 2>&1 | grep -E "fwd" || exit 1
  CFD_AMD_LIB=$lib timeout -k 10 120 python scripts/layer_bench.py --net DeltaP --batch 128 --modes fwd --only g4,g6 --bias 1 2>&1 | grep -E "fwd" || exit 1
done
