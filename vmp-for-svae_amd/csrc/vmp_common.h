// Shared host/device helpers for libvmp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/vmp_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace vmp {

constexpr int WAVE = 64;

// ---- error plumbing (thread-local message) ------------------------------------------------
void set_error(const char* fmt, ...);
int  check_launch(const char* what);

// ---- geometry of the mixture kernels -----------------------------------------------------------------
template <int D>
struct Geo {
    static constexpr int TRI   = D * (D + 1) / 2;
    static constexpr int F     = 1 + D + TRI;        // MFMA feature columns  [1 | x_d | x_d x_e (d<=e)]
    static constexpr int FT    = (F + 15) / 16;      // 16-wide feature tiles
    static constexpr int PACK  = D + TRI + 4;        // E-step parameter words per component
    static constexpr int SW    = 2 + D + D * D;      // public stats words per component
    static constexpr int PF    = F + 1;              // partial words per component: features + Nk
    static constexpr int XROWS = D + 2;              // LDS rows: x_0..x_{D-1}, ones, zeros
};

inline int pack_words(int D)  { return D + D * (D + 1) / 2 + 4; }
inline int stats_words(int D) { return 2 + D + D * D; }
inline int partial_words(int D) { return 1 + D + D * (D + 1) / 2 + 1; }

}  // namespace vmp
