// Row kernel of the fp16 memory snapshot (create_implicit_memory, custom_rcnn.py:762-774, + the fp16 cast of
// timm.py:147): out[row] = half(obs > 1 ? mem[row] / obs : mem[row]), one wave per row, 8 channels per lane.  Shared by the
// stand-alone incremental normalise (memory_read.hip) and by the observation-counter kernel of the memory write (memory.hip),
// which refreshes the rows it touches in the same pass.
#pragma once
#include "eod_common.h"
#include <hip/hip_fp16.h>

// `bal`: bit b set = cell (g << 6) + b of the 64-cell group g is to be written by THIS wave; obs_of(b) = its observation count
// (wave-uniform).  Up to 4 rows per step: their loads are independent and in flight together.
template <class ObsOf>
__device__ __forceinline__ void eod_snapshot_rows(const float* __restrict__ mem, __half* __restrict__ out, int g, unsigned long long bal,
                                                  int lane, ObsOf obs_of) {
  while (bal) {
    int cells[4], bits[4];
    int nb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (bal) {
        bits[j] = (int)__ffsll((long long)bal) - 1;
        bal &= bal - 1;
        ++nb;
      } else {
        bits[j] = bits[0];
      }
      cells[j] = (g << 6) + bits[j];
    }
    f32x4 a[4], b[4];
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* m = mem + (size_t)cells[j] * 512 + lane * 8;
      a[j] = *reinterpret_cast<const f32x4*>(m);
      b[j] = *reinterpret_cast<const f32x4*>(m + 4);
      o[j] = obs_of(bits[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < nb) {
        f32x4 x = a[j], y = b[j];
        if (o[j] > 1.0f) {
          x.x = __fdiv_rn(x.x, o[j]); x.y = __fdiv_rn(x.y, o[j]); x.z = __fdiv_rn(x.z, o[j]); x.w = __fdiv_rn(x.w, o[j]);
          y.x = __fdiv_rn(y.x, o[j]); y.y = __fdiv_rn(y.y, o[j]); y.z = __fdiv_rn(y.z, o[j]); y.w = __fdiv_rn(y.w, o[j]);
        }
        __half2 h0 = __floats2half2_rn(x.x, x.y), h1 = __floats2half2_rn(x.z, x.w);
        __half2 h2 = __floats2half2_rn(y.x, y.y), h3 = __floats2half2_rn(y.z, y.w);
        uint4 pk;
        pk.x = *reinterpret_cast<unsigned*>(&h0);
        pk.y = *reinterpret_cast<unsigned*>(&h1);
        pk.z = *reinterpret_cast<unsigned*>(&h2);
        pk.w = *reinterpret_cast<unsigned*>(&h3);
        *reinterpret_cast<uint4*>(out + (size_t)cells[j] * 512 + lane * 8) = pk;
      }
    }
  }
}
