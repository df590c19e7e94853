run() { env "$@" timeout 600 python tools/bench_configs.py cfg4:512 cfgT:1024:10 2>&1 | grep -v "amdgpu.ids\|Warn" | sed 's/.*"config": "\([^:,]*\).*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1 \2/' | tr '\n' ' '; echo; }
for rep in 1 2; do
echo "new defaults: $(run X=1)"
echo "old: $(run CDX_GEMM_SMALL_TILE_BELOW=520 CDX_GEMM_XCD_ORDER=0)"
echo "only xcd=1: $(run CDX_GEMM_SMALL_TILE_BELOW=520 CDX_GEMM_XCD_ORDER=1)"
echo "only 256: $(run CDX_GEMM_XCD_ORDER=0)"
done
strings cleandiffuser_amd/csrc/libcdx.so | grep -c "CDX_GEMM_XCD_ORDER"
