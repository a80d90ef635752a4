// vtab.h -- EXPERIMENT (round 6): the 16384-slot position table of deflate-fast.mbt:95-117 kept in
// 128 VGPRs of one wavefront (32 KiB: slot h -> register h >> 7, lane h & 63, half (h >> 6) & 1), owned by inline
// asm: v[96:223].  The compiler never sees them as values: every statement lists them as clobbers (which also makes
// the kernel descriptor allocate them) and the kernel is built with amdgpu_num_vgpr(96) so that its own values stay below.
//   vt_init            all slots 0
//   vt_gather          every lane reads the dword of its slot: 128 x {v_cmpx_le_u32 (exec = lanes whose register is
//                      >= i), ds_bpermute_b32 from register i}: a lane keeps the value of the last pass it was active
//                      in, which is its own register's; the LDS crossbar moves the data, no LDS memory is touched
//   vt_scatter_half    the lanes of a mask write their 16-bit value, one after the other: v_readlane the lane's
//                      {register, target lane, value}, exec = the target lane, s_set_gpr_idx_on (VGPR index mode:
//                      SRC2 and DST relative) + v_bfi_b32 into the table register
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define VT_CLOBBERS "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223"

__device__ __forceinline__ void vt_init() {
  asm volatile(
      ".set vt_i, 0\n\t"
      ".rept 128\n\t"
      "v_mov_b32 v[96+vt_i], 0\n\t"
      ".set vt_i, vt_i+1\n\t"
      ".endr\n\t" ::: VT_CLOBBERS);
}

// r = register of my slot (h >> 7), addr4 = 4 * lane of my slot: the dword that holds my slot.
// (ds_bpermute returns 0 from a source lane that is not in exec: the crossbar pass runs with every lane active and
// each lane keeps the pass of its own register -- groups of four passes in flight while the previous four are selected)
#define VT_BP4(t0, t1, t2, t3, base)                                \
  "ds_bpermute_b32 %[" #t0 "], %[addr], v[96+vt_i+" #base "+0]\n\t" \
  "ds_bpermute_b32 %[" #t1 "], %[addr], v[96+vt_i+" #base "+1]\n\t" \
  "ds_bpermute_b32 %[" #t2 "], %[addr], v[96+vt_i+" #base "+2]\n\t" \
  "ds_bpermute_b32 %[" #t3 "], %[addr], v[96+vt_i+" #base "+3]\n\t"
#define VT_SEL4(t0, t1, t2, t3, base)                               \
  "v_cmp_eq_u32 vcc, vt_i+" #base "+0, %[r]\n\t"                    \
  "v_cndmask_b32 %[old], %[old], %[" #t0 "], vcc\n\t"               \
  "v_cmp_eq_u32 vcc, vt_i+" #base "+1, %[r]\n\t"                    \
  "v_cndmask_b32 %[old], %[old], %[" #t1 "], vcc\n\t"               \
  "v_cmp_eq_u32 vcc, vt_i+" #base "+2, %[r]\n\t"                    \
  "v_cndmask_b32 %[old], %[old], %[" #t2 "], vcc\n\t"               \
  "v_cmp_eq_u32 vcc, vt_i+" #base "+3, %[r]\n\t"                    \
  "v_cndmask_b32 %[old], %[old], %[" #t3 "], vcc\n\t"
__device__ __forceinline__ uint32_t vt_gather(uint32_t r, uint32_t addr4) {
  uint32_t old = 0, a0, a1, a2, a3, b0, b1, b2, b3;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"  // (a scalar load still in flight would return out of order: the counts below are LDS only)
      ".set vt_i, 0\n\t"
      VT_BP4(a0, a1, a2, a3, 0)
      ".rept 15\n\t"
      VT_BP4(b0, b1, b2, b3, 4)
      "s_waitcnt lgkmcnt(4)\n\t"
      VT_SEL4(a0, a1, a2, a3, 0)
      VT_BP4(a0, a1, a2, a3, 8)
      "s_waitcnt lgkmcnt(4)\n\t"
      VT_SEL4(b0, b1, b2, b3, 4)
      ".set vt_i, vt_i+8\n\t"
      ".endr\n\t"
      VT_BP4(b0, b1, b2, b3, 4)
      "s_waitcnt lgkmcnt(4)\n\t"
      VT_SEL4(a0, a1, a2, a3, 0)
      "s_waitcnt lgkmcnt(0)\n\t"
      VT_SEL4(b0, b1, b2, b3, 4)
      : [old] "+v"(old), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1),
        [b2] "=&v"(b2), [b3] "=&v"(b3)
      : [r] "v"(r), [addr] "v"(addr4)
      : "vcc", VT_CLOBBERS);
  return old;
}

// hw = register [7:0] | target lane << 8 | value << 16 of every lane in `m`; take = 0x0000ffff (low half) or
// 0xffff0000 (high half): the bits of the table dword that the value replaces
__device__ __forceinline__ void vt_scatter_half(uint64_t m, uint32_t hw, uint32_t take) {
  uint32_t l, x, xn, t, val, m0save;
  uint64_t save;
  asm volatile(
      "s_cmp_eq_u64 %[m], 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b32 %[m0s], m0\n\t"
      "s_ff1_i32_b64 %[l], %[m]\n\t"
      "v_readlane_b32 %[x], %[hw], %[l]\n\t"
      "s_bitset0_b64 %[m], %[l]\n\t"
      "1:\n\t"
      // the next lane's word is fetched while this one is written (m == 0: lane select -1 reads lane 63, unused)
      "s_ff1_i32_b64 %[l], %[m]\n\t"
      "s_lshr_b32 %[t], %[x], 8\n\t"
      "v_readlane_b32 %[xn], %[hw], %[l]\n\t"
      "s_pack_hh_b32_b16 %[val], %[x], %[x]\n\t"
      "s_lshl_b64 exec, 1, %[t]\n\t"
      "s_set_gpr_idx_on %[x], 0xC\n\t"
      "v_bfi_b32 v96, %[take], %[val], v96\n\t"
      "s_set_gpr_idx_off\n\t"
      "s_cmp_eq_u64 %[m], 0\n\t"
      "s_bitset0_b64 %[m], %[l]\n\t"
      "s_mov_b32 %[x], %[xn]\n\t"
      "s_cbranch_scc0 1b\n\t"
      "s_mov_b32 m0, %[m0s]\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "2:\n\t"
      : [m] "+s"(m), [l] "=&s"(l), [x] "=&s"(x), [xn] "=&s"(xn), [t] "=&s"(t), [val] "=&s"(val), [save] "=&s"(save),
        [m0s] "=&s"(m0save)
      : [hw] "v"(hw), [take] "v"(take)
      : "scc", VT_CLOBBERS);
}
