// poa_kern_tables.hip.h -- kernel classes by launch geometry.
//
// The engine has ~300 kernel instantiations (strip width x workgroup size x gap model x alignment mode x plane cell
// format, block and align-only kernels): compiled in ONE translation unit they cost three minutes of hipcc.  The classes
// are therefore instantiated in PARTS, each in a translation unit of its own (kern_*.hip: `#define SXG_KERN_PART n` and
// this header) that smoothxg_amd/build.py compiles in parallel; sxg_poa.hip only sees the declarations.  Development
// builds of a single packed class (-DSXG_DEV_ONLY_W=<w> -DSXG_DEV_ONLY_TMAX=<t>: seconds) compile sxg_poa.hip alone.
//
//   part 1   32-bit sweeps (row modes 0, 1), block and align-only kernels; the banded one-wave sweep (row mode 3)
//   part 2   packed sweep (row mode 2), block kernels, 2-byte delta plane cells, workgroups of up to 4 waves
//   part 3   ... 8 and 16 waves
//   part 4   packed sweep, block kernels, 4-byte plane cells (score sets whose deltas do not fit 16 bits), up to 4 waves
//   part 5   ... 8 and 16 waves
//   part 6   packed sweep, align-only kernels (4-byte cells)
//   part 7   packed sweep, block kernels, 2-byte cells, two-wave workgroups (TMAX = 128: with part 8 the classes that may read
//            stored rows back from a full-width plane and batch their graph phases 8 / 16 elements per thread)
//   part 8   ... one-wave workgroups (TMAX = 64: no wave-to-wave hand-over compiled in)
//   part 9   packed sweep, block kernels, 2-byte cells, three-wave workgroups (TMAX = 192)
// The 2-byte classes of up to eight waves run at EXACTLY TMAX threads, and the sweep is compiled for that thread count (1.6 % on the
// headline; the two-wave class measured 1 % slower that way and keeps reading it at run time); the 4-byte classes and the 16-wave
// ones take their thread count at run time.
#pragma once
#include "poa_kernels.hip.h"

// A launch geometry: W columns per strip, NW waves (T = 64*NW), kernel class TMAX, row mode RM
// (2 = packed sweep: two strips per lane), CB = bytes per plane cell of the packed sweep (poa_dp16.hip.h).
struct Variant {
    int W, NW, TMAX, RM;
    int CB = 4;
    bool DS = false;   // the class compiled for smoothxg's default scores (packed sweep, 2-byte cells, convex)
    int T() const { return 64 * NW; }
    int Lpad() const { return 64 * NW * W * (RM >= 2 ? 2 : 1); }
};

template <class Args> using KernelFn = void (*)(const Args);

KernelFn<BlockArgs> sxg_block_kernel_part1(const Variant& v, bool cvx, bool sw);
KernelFn<AlignArgs> sxg_align_kernel_part1(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part2(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part3(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part4(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part5(const Variant& v, bool cvx, bool sw);
KernelFn<AlignArgs> sxg_align_kernel_part6(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part7(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part8(const Variant& v, bool cvx, bool sw);
KernelFn<BlockArgs> sxg_block_kernel_part9(const Variant& v, bool cvx, bool sw);

#if defined(SXG_KERN_PART) || defined(SXG_DEV_ONLY_W)
template <int TMAX, int W, int RM, int CB = 4> static KernelFn<BlockArgs> pick_block(bool cvx, bool sw, bool ds = false) {
    if constexpr (RM == 2 && CB == 2) {
        if (cvx && ds) return sw ? poa_block_kernel<TMAX, W, true, RM, true, CB, true> : poa_block_kernel<TMAX, W, true, RM, false, CB, true>;
    }
    if (cvx) return sw ? poa_block_kernel<TMAX, W, true, RM, true, CB> : poa_block_kernel<TMAX, W, true, RM, false, CB>;
    return sw ? poa_block_kernel<TMAX, W, false, RM, true, CB> : poa_block_kernel<TMAX, W, false, RM, false, CB>;
}
template <int TMAX, int W, int CB> static KernelFn<BlockArgs> pick_block_sw(bool cvx, bool ds) {   // (packed, local alignment only)
    if constexpr (CB == 2) {
        if (cvx && ds) return poa_block_kernel<TMAX, W, true, 2, true, CB, true>;
    }
    return cvx ? poa_block_kernel<TMAX, W, true, 2, true, CB> : poa_block_kernel<TMAX, W, false, 2, true, CB>;
}
template <int TMAX, int W, int RM> static KernelFn<AlignArgs> pick_align(bool cvx, bool sw) {
    if (cvx) return sw ? poa_align_kernel<TMAX, W, true, RM, true> : poa_align_kernel<TMAX, W, true, RM, false>;
    return sw ? poa_align_kernel<TMAX, W, false, RM, true> : poa_align_kernel<TMAX, W, false, RM, false>;
}
#define SXG_PICK(FN, TM, Wd)                                             \
    do {                                                                 \
        if (v.TMAX == TM && v.W == Wd) {                                 \
            if (v.RM == 0) return FN<TM, Wd, 0>(cvx, sw);                \
            if (v.RM == 1) return FN<TM, Wd, 1>(cvx, sw);                \
        }                                                                \
    } while (0)
#define SXG_PICK16B(TM, Wd, CBv) \
    do { if (v.TMAX == TM && v.W == Wd && v.RM == 2 && v.CB == CBv) return pick_block<TM, Wd, 2, CBv>(cvx, sw, v.DS); } while (0)
// (the long classes exist for local alignment only: a global score of such lengths does not fit int16)
#define SXG_PICK16B_SW(TM, Wd, CBv) \
    do { if (v.TMAX == TM && v.W == Wd && v.RM == 2 && v.CB == CBv && sw) return pick_block_sw<TM, Wd, CBv>(cvx, v.DS); } while (0)
#define SXG_PICK16A(TM, Wd) \
    do { if (v.TMAX == TM && v.W == Wd && v.RM == 2) return pick_align<TM, Wd, 2>(cvx, sw); } while (0)
#define SXG_PICK16A_SW(TM, Wd) \
    do { if (v.TMAX == TM && v.W == Wd && v.RM == 2 && sw) return cvx ? poa_align_kernel<TM, Wd, true, 2, true> : poa_align_kernel<TM, Wd, false, 2, true>; } while (0)
#endif

#if defined(SXG_DEV_ONLY_W)
// development: ONE packed class, both cell formats, everything in the including translation unit
KernelFn<BlockArgs> sxg_block_kernel_part1(const Variant&, bool, bool) { return nullptr; }
KernelFn<AlignArgs> sxg_align_kernel_part1(const Variant&, bool, bool) { return nullptr; }
KernelFn<BlockArgs> sxg_block_kernel_part2(const Variant& v, bool cvx, bool sw) { SXG_PICK16B(SXG_DEV_ONLY_TMAX, SXG_DEV_ONLY_W, 2); return nullptr; }
KernelFn<BlockArgs> sxg_block_kernel_part3(const Variant&, bool, bool) { return nullptr; }
#ifdef SXG_DEV_CB4
KernelFn<BlockArgs> sxg_block_kernel_part4(const Variant& v, bool cvx, bool sw) { SXG_PICK16B(SXG_DEV_ONLY_TMAX, SXG_DEV_ONLY_W, 4); return nullptr; }
#else
KernelFn<BlockArgs> sxg_block_kernel_part4(const Variant&, bool, bool) { return nullptr; }
#endif
KernelFn<BlockArgs> sxg_block_kernel_part5(const Variant&, bool, bool) { return nullptr; }
KernelFn<AlignArgs> sxg_align_kernel_part6(const Variant&, bool, bool) { return nullptr; }
KernelFn<BlockArgs> sxg_block_kernel_part7(const Variant& v, bool cvx, bool sw) { return sxg_block_kernel_part2(v, cvx, sw); }
KernelFn<BlockArgs> sxg_block_kernel_part8(const Variant& v, bool cvx, bool sw) { return sxg_block_kernel_part2(v, cvx, sw); }
KernelFn<BlockArgs> sxg_block_kernel_part9(const Variant& v, bool cvx, bool sw) { return sxg_block_kernel_part2(v, cvx, sw); }
#elif defined(SXG_KERN_PART)
#if SXG_KERN_PART == 1
KernelFn<BlockArgs> sxg_block_kernel_part1(const Variant& v, bool cvx, bool sw) {
    if (v.RM == 3) {   // banded: the strip width is part of the semantics (decree B2), never merged or widened
        if (v.CB == 2 && sw) {   // (round 6) 2-byte band cells: local alignment, score sets whose delta code fits 16 bits
            if (v.W == 6) return cvx ? poa_block_kernel<64, 6, true, 3, true, 2> : poa_block_kernel<64, 6, false, 3, true, 2>;
            if (v.W == 8) return cvx ? poa_block_kernel<64, 8, true, 3, true, 2> : poa_block_kernel<64, 8, false, 3, true, 2>;
            return cvx ? poa_block_kernel<64, 11, true, 3, true, 2> : poa_block_kernel<64, 11, false, 3, true, 2>;
        }
        if (v.W == 6) return pick_block<64, 6, 3>(cvx, sw);
        if (v.W == 8) return pick_block<64, 8, 3>(cvx, sw);
        return pick_block<64, 11, 3>(cvx, sw);
    }
    SXG_PICK(pick_block, 256, 8); SXG_PICK(pick_block, 256, 12); SXG_PICK(pick_block, 256, 16);
    SXG_PICK(pick_block, 512, 8); SXG_PICK(pick_block, 512, 12); SXG_PICK(pick_block, 512, 16);
    SXG_PICK(pick_block, 1024, 8); SXG_PICK(pick_block, 1024, 12);
    return nullptr;
}
KernelFn<AlignArgs> sxg_align_kernel_part1(const Variant& v, bool cvx, bool sw) {
    SXG_PICK(pick_align, 256, 8); SXG_PICK(pick_align, 256, 12); SXG_PICK(pick_align, 256, 16);
    SXG_PICK(pick_align, 512, 8); SXG_PICK(pick_align, 512, 12); SXG_PICK(pick_align, 512, 16);
    SXG_PICK(pick_align, 1024, 8); SXG_PICK(pick_align, 1024, 12);
    return nullptr;
}
#elif SXG_KERN_PART == 2 || SXG_KERN_PART == 4
#if SXG_KERN_PART == 2
#define SXG_PART_CB 2
KernelFn<BlockArgs> sxg_block_kernel_part2(const Variant& v, bool cvx, bool sw) {
#else
#define SXG_PART_CB 4
KernelFn<BlockArgs> sxg_block_kernel_part4(const Variant& v, bool cvx, bool sw) {
#endif
    SXG_PICK16B(256, 4, SXG_PART_CB); SXG_PICK16B(256, 5, SXG_PART_CB); SXG_PICK16B(256, 6, SXG_PART_CB); SXG_PICK16B(256, 7, SXG_PART_CB);
    SXG_PICK16B(256, 8, SXG_PART_CB); SXG_PICK16B(256, 9, SXG_PART_CB); SXG_PICK16B(256, 10, SXG_PART_CB);
    SXG_PICK16B(256, 11, SXG_PART_CB); SXG_PICK16B(256, 12, SXG_PART_CB);
    return nullptr;
}
#elif SXG_KERN_PART == 3 || SXG_KERN_PART == 5
#if SXG_KERN_PART == 3
#define SXG_PART_CB 2
KernelFn<BlockArgs> sxg_block_kernel_part3(const Variant& v, bool cvx, bool sw) {
#else
#define SXG_PART_CB 4
KernelFn<BlockArgs> sxg_block_kernel_part5(const Variant& v, bool cvx, bool sw) {
#endif
    SXG_PICK16B(512, 8, SXG_PART_CB); SXG_PICK16B(512, 9, SXG_PART_CB); SXG_PICK16B(512, 10, SXG_PART_CB);
    SXG_PICK16B(512, 11, SXG_PART_CB); SXG_PICK16B(512, 12, SXG_PART_CB);
    SXG_PICK16B(1024, 8, SXG_PART_CB);
    SXG_PICK16B_SW(1024, 10, SXG_PART_CB); SXG_PICK16B_SW(1024, 12, SXG_PART_CB); SXG_PICK16B_SW(1024, 13, SXG_PART_CB);
    return nullptr;
}
#elif SXG_KERN_PART == 7
KernelFn<BlockArgs> sxg_block_kernel_part7(const Variant& v, bool cvx, bool sw) {
    SXG_PICK16B(128, 4, 2); SXG_PICK16B(128, 5, 2); SXG_PICK16B(128, 6, 2); SXG_PICK16B(128, 7, 2);
    SXG_PICK16B(128, 8, 2); SXG_PICK16B(128, 9, 2); SXG_PICK16B(128, 10, 2);
    SXG_PICK16B(128, 11, 2); SXG_PICK16B(128, 12, 2);
    return nullptr;
}
#elif SXG_KERN_PART == 8
KernelFn<BlockArgs> sxg_block_kernel_part8(const Variant& v, bool cvx, bool sw) {
    SXG_PICK16B(64, 4, 2); SXG_PICK16B(64, 5, 2); SXG_PICK16B(64, 6, 2); SXG_PICK16B(64, 7, 2);
    SXG_PICK16B(64, 8, 2); SXG_PICK16B(64, 9, 2); SXG_PICK16B(64, 10, 2);
    SXG_PICK16B(64, 11, 2); SXG_PICK16B(64, 12, 2);
    return nullptr;
}
#elif SXG_KERN_PART == 9
KernelFn<BlockArgs> sxg_block_kernel_part9(const Variant& v, bool cvx, bool sw) {
    SXG_PICK16B(192, 4, 2); SXG_PICK16B(192, 5, 2); SXG_PICK16B(192, 6, 2); SXG_PICK16B(192, 7, 2);
    SXG_PICK16B(192, 8, 2); SXG_PICK16B(192, 9, 2); SXG_PICK16B(192, 10, 2);
    SXG_PICK16B(192, 11, 2); SXG_PICK16B(192, 12, 2);
    return nullptr;
}
#elif SXG_KERN_PART == 6
KernelFn<AlignArgs> sxg_align_kernel_part6(const Variant& v, bool cvx, bool sw) {
    SXG_PICK16A(256, 4); SXG_PICK16A(256, 5); SXG_PICK16A(256, 6); SXG_PICK16A(256, 7);
    SXG_PICK16A(256, 8); SXG_PICK16A(256, 9); SXG_PICK16A(256, 10);
    SXG_PICK16A(256, 11); SXG_PICK16A(256, 12);
    SXG_PICK16A(512, 8); SXG_PICK16A(512, 9); SXG_PICK16A(512, 10);
    SXG_PICK16A(512, 11); SXG_PICK16A(512, 12);
    SXG_PICK16A(1024, 8);
    SXG_PICK16A_SW(1024, 10); SXG_PICK16A_SW(1024, 12); SXG_PICK16A_SW(1024, 13);
    return nullptr;
}
#endif
#endif

// the kernel class of a geometry (nullptr: none built)
static inline KernelFn<BlockArgs> block_kernel(const Variant& v, bool cvx, bool sw) {
    if (v.RM != 2) return sxg_block_kernel_part1(v, cvx, sw);
    if (v.CB == 2) return v.TMAX <= 64 ? sxg_block_kernel_part8(v, cvx, sw) : v.TMAX <= 128 ? sxg_block_kernel_part7(v, cvx, sw) : v.TMAX <= 192 ? sxg_block_kernel_part9(v, cvx, sw) : (v.TMAX <= 256 ? sxg_block_kernel_part2(v, cvx, sw) : sxg_block_kernel_part3(v, cvx, sw));
    Variant u = v;
    if (u.TMAX < 256) u.TMAX = 256;   // (4-byte cells: no class of its own for one and two waves)
    return u.TMAX <= 256 ? sxg_block_kernel_part4(u, cvx, sw) : sxg_block_kernel_part5(u, cvx, sw);
}
static inline KernelFn<AlignArgs> align_kernel(const Variant& v, bool cvx, bool sw) {
    Variant u = v;
    if (u.RM == 2 && u.TMAX < 256) u.TMAX = 256;
    return u.RM != 2 ? sxg_align_kernel_part1(u, cvx, sw) : sxg_align_kernel_part6(u, cvx, sw);
}
