// Shared by conv2d.hip (dense implicit-GEMM kernels) and conv2d_sparse.hip (structurally sparse input).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IG_T 128
#define IG_LD 132

struct IgParams {
    const float *A;   // FWD: W [Cout][Cin*9]; DGRAD: Wd class slab [Cin][Cout*nt]; WGRAD: gY
    const float *Bm;  // FWD/WGRAD: X; DGRAD: gY
    float *D;         // FWD: Y; DGRAD: gX; WGRAD: partial slab [split][Cout][Cin*9]
    int Bn, Cin, H, W, Cout, Ho, Wo;
    int Mi, Nj, Kd;   // GEMM extents
    // DGRAD class description
    int py, px, nky, nkx, Hc, Wc;
    int ky0, kystep, kx0, kxstep;
    // WGRAD split
    int k_chunk;
};

enum { IG_FWD = 0, IG_DGRAD = 1, IG_WGRAD = 2 };

// up to four problems in one launch (the stride-2 parity classes of a data gradient, longest reduction first)
struct IgClasses {
    IgParams c[4];
    int first[5];            // first workgroup of every class; first[n] = grid size
    int n;
    int row_tiles[4];        // row tiles of a pixel tile (next to each other in the grid)
    long long list_off[4];   // sparse data gradient: the class's tile lists inside the activity buffer (int32 units)
};

#define C2_KB 32  // reduction depth of one LDS stage (FWD / DGRAD)

// Wd = the data-gradient weight slabs of the four stride-2 parity classes of a 3x3 kernel W [Cout][Cin][3][3], in class
// order (9 Cin Cout floats), one launch (conv2d.hip)
void lidog_launch_repack_dgrad_all(const float *W, int Cin, int Cout, float *Wd, hipStream_t st);
