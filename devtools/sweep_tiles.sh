for e in "" "ASLP_GEMM_TILE_NT=207" "ASLP_GEMM_TILE_NT=213" "ASLP_GEMM_TILE_NT=208" "ASLP_GEMM_TILE_NN=213" "ASLP_GEMM_TILE_TN=211" "ASLP_GEMM_TILE_TN=208" "ASLP_GEMM_TILE_TN=207"; do
  echo "== $e"; env $e timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k:(round(v['avg_us'],1),round(v['tflops'],1)) for k,v in d['gemm_all']['variants'].items()})"
done
