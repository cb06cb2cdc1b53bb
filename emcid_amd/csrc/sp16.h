// Split-fp16 rows (include/emcid_hip.h, "split fp16"): helpers shared by the projection kernel (gemm_sp16.hip) and by the
// producers that write their result straight as planes (LayerNorm, tree attention: attention.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace emcid {

typedef _Float16 sp_v4h __attribute__((ext_vector_type(4)));

// 2^e for a row whose largest magnitude (or a bound on it) is amax: amax 2^e in [2^14, 2^15) (e clamped to +-120; zero /
// subnormal rows get the clamp, inf / nan rows propagate through hi).
__device__ __forceinline__ void sp_scale_of(float amax, float& s, float& inv) {
    int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
    ex = ex == 0 ? 1 : (ex == 255 ? 254 : ex);
    int e = 141 - ex;                                          // 14 - (ex - 127)
    e = e > 120 ? 120 : (e < -120 ? -120 : e);
    s = __uint_as_float((unsigned)(e + 127) << 23);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
}

// four consecutive elements x[col .. col + 3] (col % 4 == 0) of a row, scaled by s, into the row's planes:
// group col / 8 is [hi x 8 (16 bytes)][lo x 8 (16 bytes)], the four elements are its first or second half
__device__ __forceinline__ void sp_store4(uint32_t* row_planes, int col, float x0, float x1, float x2, float x3, float s) {
    sp_v4h h, l;
    const float t0 = x0 * s, t1 = x1 * s, t2 = x2 * s, t3 = x3 * s;
    h[0] = (_Float16)t0; h[1] = (_Float16)t1; h[2] = (_Float16)t2; h[3] = (_Float16)t3;
    l[0] = (_Float16)(t0 - (float)h[0]); l[1] = (_Float16)(t1 - (float)h[1]);
    l[2] = (_Float16)(t2 - (float)h[2]); l[3] = (_Float16)(t3 - (float)h[3]);
    unsigned char* dst = reinterpret_cast<unsigned char*>(row_planes) + (col >> 3) * 32 + ((col >> 2) & 1) * 8;
    *reinterpret_cast<sp_v4h*>(dst) = h;
    *reinterpret_cast<sp_v4h*>(dst + 16) = l;
}

}  // namespace emcid
