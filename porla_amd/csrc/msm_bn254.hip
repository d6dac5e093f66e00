// Bn254G1 instantiation of the bucket MSM (kernels + launch sequence); see msm.hip.h / msm_impl.hip.h.
#include "msm_impl.hip.h"
#include "fixed_base_impl.hip.h"

namespace porla {
template int msm_device<Bn254G1>(const uint8_t*, const uint8_t*, size_t, hipStream_t, XYZZ<Bn254Fp>*);
template int msm_host<Bn254G1>(const uint8_t*, const uint8_t*, size_t, XYZZ<Bn254Fp>*);
template int msm_host_multi<Bn254G1>(const uint8_t*, const uint8_t*, size_t, int, int, XYZZ<Bn254Fp>*);
template int msm_pair_device<Bn254G1>(const uint8_t*, const uint8_t*, const uint8_t*, size_t, hipStream_t, XYZZ<Bn254Fp>*, XYZZ<Bn254Fp>*);
template int msm_pair_gather_device<Bn254G1>(const uint8_t*, const uint8_t*, const uint64_t*, const uint32_t*, size_t, hipStream_t, XYZZ<Bn254Fp>*, XYZZ<Bn254Fp>*);
template int msm_pair_gather_begin<Bn254G1>(int, const uint8_t*, const uint8_t*, const uint64_t*, const uint32_t*, size_t, hipStream_t);
template int msm_pair_end<Bn254G1>(int, XYZZ<Bn254Fp>*, XYZZ<Bn254Fp>*);
template int msm_pair_host<Bn254G1>(const uint8_t*, const uint8_t*, const uint8_t*, size_t, XYZZ<Bn254Fp>*, XYZZ<Bn254Fp>*);
template int msm_device_begin<Bn254G1>(int, const uint8_t*, const uint8_t*, size_t, hipStream_t);
template int msm_device_end<Bn254G1>(int, XYZZ<Bn254Fp>*);
template struct FixedBase<Bn254G1>;
}  // namespace porla
