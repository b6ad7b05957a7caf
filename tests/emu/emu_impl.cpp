// Instantiates the fiber scheduler of the CPU emulation shim (TEST INFRASTRUCTURE ONLY).
#define CSN_EMU_IMPL
#include "hip_cpu_shim.h"

#ifdef CSN_EMU_LANES
#include "csn_device.h"   // kinds 11 / 12: the kernels' own helpers on the shim's lane operations
// One wave executing ONE cross-lane instruction of the lane-exact mode, operands and results per lane -- the twin of
// tests/emu/lane_probe.hip, which runs the real instruction on the GPU (tests/test_gpu_lane_ops.py compares the two bit for bit;
// tests/test_emu_lanes.py compares this one with the instructions' matrix definitions).
// a, b: [64][16] bytes; acc_in / acc_out: [64][16] floats.  kind as csn_emu_lane_ops; kind + 16: under a lane mask (lanes whose
// first dword of b is zero do not execute the instruction and return their `a` unchanged: DPP kinds and readfirstlane only).
extern "C" int csn_emu_lane_probe(int kind, const unsigned char* a, const unsigned char* b, const float* acc_in, float* acc_out) {
  int bad = 0;
  csn_emu::launch(dim3(1), dim3(64), 0, [&]() {
    const unsigned lane = threadIdx.x;
    const unsigned char* la = a + 16 * lane;
    const unsigned char* lb = b + 16 * lane;
    float d[16];
    for (int i = 0; i < 16; ++i) d[i] = acc_in[16 * lane + i];
    float fa, fb;
    unsigned ua, ub;
    std::memcpy(&fa, la, 4); std::memcpy(&fb, lb, 4);
    std::memcpy(&ua, la, 4); std::memcpy(&ub, lb, 4);
    const bool masked = kind >= 16;
    unsigned r = ua;
    switch (kind & 15) {
      case 1: csn_emu::lanes_mfma_f32_4x4x1(fa, fb, d); break;
      case 2: csn_emu::lanes_mfma_f32_4x4x4_bf16(la, lb, d); break;
      case 3: csn_emu::lanes_mfma_f32_16x16x4_f32(fa, fb, d); break;
      case 4: csn_emu::lanes_mfma_f32_32x32x16_bf16(la, lb, d); break;
      case 5: csn_emu::lanes_mfma_f32_16x16x32_bf16(la, lb, d); break;
      case 6: if (!masked || ub) r = csn_emu::lanes_dpp_wave_shr1(ua); break;
      case 7: if (!masked || ub) r = csn_emu::lanes_dpp_wave_shl1(ua); break;
      case 8: if (!masked || ub) r = csn_emu::lanes_readfirstlane(ua); break;
      case 10: {   // 64-bit shuffle: a = the value (two dwords), b = the xor mask
        unsigned long long v; std::memcpy(&v, la, 8);
        v = csn_emu::lanes_shfl_xor64(v, (int)(ub & 63));
        std::memcpy(&d[0], &v, 8);
        break;
      }
      case 11: {   // csn_lane_xor_f32: a = the float, b = X
        float o = fa;
        switch (ub) {
          case 1: o = csn_lane_xor_f32<1>(fa); break;
          case 2: o = csn_lane_xor_f32<2>(fa); break;
          case 4: o = csn_lane_xor_f32<4>(fa); break;
          case 8: o = csn_lane_xor_f32<8>(fa); break;
          case 16: o = csn_lane_xor_f32<16>(fa); break;
          case 32: o = csn_lane_xor_f32<32>(fa); break;
          default: bad = 1;
        }
        d[0] = o;
        break;
      }
      case 12: {   // csn_wave_reduce_scatter8 of acc[0 .. 7]
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = d[i];
        d[0] = csn_wave_reduce_scatter8(v, (int)lane);
        d[1] = (float)csn_rs8_index((int)lane);
        break;
      }
      default: bad = 1;
    }
    if ((kind & 15) >= 6 && (kind & 15) <= 8) std::memcpy(&d[0], &r, 4);
    for (int i = 0; i < 16; ++i) acc_out[16 * lane + i] = d[i];
  });
  return bad;
}
#endif
