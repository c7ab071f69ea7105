// trl_tables.h -- piece geometry, SRS+ kick lists and policy-plane maps for the device code,
// bit-packed (own layout).  The data restates const.py:82-136 (policy planes),
// const.py:191-235 (wallkicks / i_wallkicks) and const.py:238-281 (mino_coords_dict).
#pragma once
#include <stdint.h>

namespace trl {

enum : int { PZ = 0, PL = 1, PO = 2, PS = 3, PI = 4, PJ = 5, PT = 6, P_NONE = 7 };   // "ZLOSIJT", const.py:68

// ---- minos: 4 x (cx:2, cy:2) per (type, rot) -------------------------------------------------
constexpr uint16_t mp(int x0, int y0, int x1, int y1, int x2, int y2, int x3, int y3) {
    return (uint16_t)((x0 | (y0 << 2)) | ((x1 | (y1 << 2)) << 4) | ((x2 | (y2 << 2)) << 8) | ((x3 | (y3 << 2)) << 12));
}
static __device__ __constant__ const uint16_t C_MINO[7 * 4] = {
    /* Z */ mp(0,0,1,0,1,1,2,1), mp(1,1,1,2,2,0,2,1), mp(0,1,1,1,1,2,2,2), mp(0,1,0,2,1,0,1,1),
    /* L */ mp(0,1,1,1,2,0,2,1), mp(1,0,1,1,1,2,2,2), mp(0,1,0,2,1,1,2,1), mp(0,0,1,0,1,1,1,2),
    /* O */ mp(0,0,0,1,1,0,1,1), mp(0,0,0,1,1,0,1,1), mp(0,0,0,1,1,0,1,1), mp(0,0,0,1,1,0,1,1),
    /* S */ mp(0,1,1,0,1,1,2,0), mp(1,0,1,1,2,1,2,2), mp(0,2,1,1,1,2,2,1), mp(0,0,0,1,1,1,1,2),
    /* I */ mp(0,1,1,1,2,1,3,1), mp(2,0,2,1,2,2,2,3), mp(0,2,1,2,2,2,3,2), mp(1,0,1,1,1,2,1,3),
    /* J */ mp(0,0,0,1,1,1,2,1), mp(1,0,1,1,1,2,2,0), mp(0,1,1,1,2,1,2,2), mp(0,2,1,0,1,1,1,2),
    /* T */ mp(0,1,1,0,1,1,2,1), mp(1,0,1,1,1,2,2,1), mp(0,1,1,1,1,2,2,1), mp(0,1,1,0,1,1,1,2),
};

// ---- kicks: per (table, from-rotation, direction-1): n in bits 60..62, kick i = (kx+2):3 | (ky+2):3 at bit 6i
constexpr uint64_t kk(int n, int x0 = 0, int y0 = 0, int x1 = 0, int y1 = 0, int x2 = 0, int y2 = 0,
                      int x3 = 0, int y3 = 0, int x4 = 0, int y4 = 0, int x5 = 0, int y5 = 0) {
    return ((uint64_t)n << 60)
         | ((uint64_t)((x0 + 2) | ((y0 + 2) << 3)) << 0)  | ((uint64_t)((x1 + 2) | ((y1 + 2) << 3)) << 6)
         | ((uint64_t)((x2 + 2) | ((y2 + 2) << 3)) << 12) | ((uint64_t)((x3 + 2) | ((y3 + 2) << 3)) << 18)
         | ((uint64_t)((x4 + 2) | ((y4 + 2) << 3)) << 24) | ((uint64_t)((x5 + 2) | ((y5 + 2) << 3)) << 30);
}
// index [(isI * 4 + from) * 3 + (dir - 1)], to = (from + dir) % 4
static __device__ __constant__ const uint64_t C_KICK[2 * 4 * 3] = {
    // wallkicks (all pieces but I)
    /* 0->1 */ kk(5, 0,0, -1,0, -1,1, 0,-2, -1,-2),
    /* 0->2 */ kk(6, 0,0, 0,1, 1,1, -1,1, 1,0, -1,0),
    /* 0->3 */ kk(5, 0,0, 1,0, 1,1, 0,-2, 1,-2),
    /* 1->2 */ kk(5, 0,0, 1,0, 1,-1, 0,2, 1,2),
    /* 1->3 */ kk(6, 0,0, 1,0, 1,2, 1,1, 0,2, 0,1),
    /* 1->0 */ kk(5, 0,0, 1,0, 1,-1, 0,2, 1,2),
    /* 2->3 */ kk(5, 0,0, 1,0, 1,1, 0,-2, 1,-2),
    /* 2->0 */ kk(6, 0,0, 0,-1, -1,-1, 1,-1, -1,0, 1,0),
    /* 2->1 */ kk(5, 0,0, -1,0, -1,1, 0,-2, -1,-2),
    /* 3->0 */ kk(5, 0,0, -1,0, -1,-1, 0,2, -1,2),
    /* 3->1 */ kk(6, 0,0, -1,0, -1,2, -1,1, 0,2, 0,1),
    /* 3->2 */ kk(5, 0,0, -1,0, -1,-1, 0,2, -1,2),
    // i_wallkicks
    /* 0->1 */ kk(5, 0,0, -2,0, 1,0, -2,-1, 1,2),
    /* 0->2 */ kk(2, 0,0, 0,1),
    /* 0->3 */ kk(5, 0,0, -1,0, 2,0, -1,2, 2,-1),
    /* 1->2 */ kk(5, 0,0, -1,0, 2,0, -1,2, 2,-1),
    /* 1->3 */ kk(2, 0,0, 1,0),
    /* 1->0 */ kk(5, 0,0, 2,0, -1,0, 2,1, -1,-2),
    /* 2->3 */ kk(5, 0,0, 2,0, -1,0, 2,1, -1,-2),
    /* 2->0 */ kk(2, 0,0, 0,-1),
    /* 2->1 */ kk(5, 0,0, 1,0, -2,0, 1,-2, -2,1),
    /* 3->0 */ kk(5, 0,0, 1,0, -2,0, 1,-2, -2,1),
    /* 3->1 */ kk(2, 0,0, -1,0),
    /* 3->2 */ kk(5, 0,0, -2,0, 1,0, -2,-1, 1,2),
};

// ---- policy planes (const.py:82-136) ------------------------------------------------------------
// plane -> type | rot << 3 | tspin << 5
constexpr uint8_t pp(int t, int r, int s) { return (uint8_t)(t | (r << 3) | (s << 5)); }
static __device__ __constant__ const uint8_t C_PLANE[27] = {
    pp(PO,0,0),
    pp(PZ,0,0), pp(PZ,1,0), pp(PS,0,0), pp(PS,1,0), pp(PI,0,0), pp(PI,1,0),
    pp(PL,0,0), pp(PL,1,0), pp(PL,2,0), pp(PL,3,0), pp(PJ,0,0), pp(PJ,1,0), pp(PJ,2,0), pp(PJ,3,0),
    pp(PT,0,0), pp(PT,1,0), pp(PT,2,0), pp(PT,3,0), pp(PT,0,1), pp(PT,1,1), pp(PT,2,1), pp(PT,3,1),
    pp(PT,0,2), pp(PT,1,2), pp(PT,2,2), pp(PT,3,2),
};
// first plane / number of distinct rotations / bounding-box height (len(piece_dict[type])) per type
static __device__ __constant__ const uint8_t C_PLANE_BASE[7] = {1, 7, 0, 3, 5, 11, 15};
static __device__ __constant__ const uint8_t C_NROT[7]       = {2, 4, 1, 2, 2, 4, 4};
static __device__ __constant__ const uint8_t C_BOX[7]        = {3, 3, 2, 3, 4, 3, 3};

}  // namespace trl
