# does the row-lane self test catch a mis-counted wait?  libsigma_hip_brokenwait.so = the product build with
# rl_dma_wait_keep<3 + 2 NS + 3> in scan_bwdr.hip (the backward reads u / delta / dout before the LDS-DMA has landed)
for v in "" _brokenwait; do
  SIGMA_HIP_LIB=$PWD/sigma_amd/lib/libsigma_hip$v.so python - <<PY
from sigma_amd import _capi
lib = _capi.load()
rcs = [lib.sigma_scan_rowlane_selftest(None) for _ in range(3)]
print("libsigma_hip$v:", rcs, _capi.last_error() if any(rcs) else "pass")
PY
done
