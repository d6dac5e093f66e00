// Secp256k1G instantiation of the bucket MSM (kernels + launch sequence); see msm.hip.h / msm_impl.hip.h.
#include "msm_impl.hip.h"
#include "fixed_base_impl.hip.h"

namespace porla {
template int msm_device<Secp256k1G>(const uint8_t*, const uint8_t*, size_t, hipStream_t, XYZZ<Secp256k1Fp>*);
template int msm_host<Secp256k1G>(const uint8_t*, const uint8_t*, size_t, XYZZ<Secp256k1Fp>*);
template int msm_host_multi<Secp256k1G>(const uint8_t*, const uint8_t*, size_t, int, int, XYZZ<Secp256k1Fp>*);
template int msm_pair_device<Secp256k1G>(const uint8_t*, const uint8_t*, const uint8_t*, size_t, hipStream_t, XYZZ<Secp256k1Fp>*, XYZZ<Secp256k1Fp>*);
template int msm_pair_gather_device<Secp256k1G>(const uint8_t*, const uint8_t*, const uint64_t*, const uint32_t*, size_t, hipStream_t, XYZZ<Secp256k1Fp>*, XYZZ<Secp256k1Fp>*);
template int msm_pair_gather_begin<Secp256k1G>(int, const uint8_t*, const uint8_t*, const uint64_t*, const uint32_t*, size_t, hipStream_t);
template int msm_pair_end<Secp256k1G>(int, XYZZ<Secp256k1Fp>*, XYZZ<Secp256k1Fp>*);
template int msm_pair_host<Secp256k1G>(const uint8_t*, const uint8_t*, const uint8_t*, size_t, XYZZ<Secp256k1Fp>*, XYZZ<Secp256k1Fp>*);
template int msm_device_begin<Secp256k1G>(int, const uint8_t*, const uint8_t*, size_t, hipStream_t);
template int msm_device_end<Secp256k1G>(int, XYZZ<Secp256k1Fp>*);
template struct FixedBase<Secp256k1G>;
}  // namespace porla
