// gel_eval_kernel.h -- the fused evaluation kernel (included by gel_kernels.hip).
//
// One wavefront = one (decision vector b, phase, 64-node chunk) work item; one
// lane = one collocation node.  The phase is wave-uniform: its parameters sit in
// SGPRs, every phase-type branch (air / NoAir, hold, engine off) is a scalar
// branch.  Forms (template parameters):
//   throughput (MFMA, !SPLIT): a workgroup = one work item x four decision vectors, D.X formed together on the
//     matrix pipe from state rows staged in LDS -- at once for phases below 68 nodes, in double-buffered 44-row
//     slabs above; PACK (every phase <= 32 nodes): two vectors per wavefront, eight per workgroup;
//   latency (SPLIT, a handful of vectors): four wavefronts per work item, one position sweep each;
//   !MFMA: wavefront dot-products with scalar (SMEM, broadcast) X loads -- selectable, not the default.
//
// Memory discipline: vector-memory operations retire in order (one vmcnt per
// wave), so a global load issued after a store waits for that store to drain to
// HBM.  Therefore EVERY global load happens before the first store: the D.X
// product runs first, the groups that need no velocity RHS (position Jacobian
// entries, the whole quaternion group) are finished right after it while their
// inputs are in registers, and what the velocity group needs late is parked in
// LDS (its own lgkmcnt) instead of being re-read or kept in VGPRs.
//
// Register discipline (fp64 = 2 VGPRs per value, 64-wide; <= 128 VGPRs -> 4 waves
// per SIMD): the heavy chain (geodetic -> atmosphere -> wind -> Earth-angle
// quaternion -> aero) has ONE instance, in a loop whose LAST trip is the centre
// point and leaves through `break`; across a trip only the node position, the
// Earth angle and 1/m stay in registers -- velocity and thrust direction are
// re-formed from the park, position-sweep results wait in the park, and centre
// values that only a later sweep block needs are parked as soon as they exist.
#pragma once

namespace gel {

// per-lane LDS park, one region per wavefront ([slot][lane]: consecutive lanes hit consecutive banks).
// The MFMA D.X path first uses the wave's region as the 64 x 17 staging tile of its result.
// Q, V, DJJ: inputs the velocity sweeps re-read; LV: D.X row of the velocity defect; FP: results of the
// position sweeps waiting for the centre value.  19 slots = 9.5 KB per wavefront: 16 wavefronts per CU fit
// the 160 KB LDS.
enum ParkSlot { PK_Q0 = 0, PK_Q1, PK_Q2, PK_Q3, PK_V0, PK_V1, PK_V2, PK_DJJ, PK_LV0, PK_LV1, PK_LV2,
                PK_FP0, PK_FP1, PK_FP2, PK_FP3, PK_FP4, PK_FP5, PK_FP6, PK_FP7, PK_COUNT };
constexpr int kStageLd = 17;                     // padded row of the staging tile: conflict-free row reads
constexpr int kWaveLds = PK_COUNT * 64;          // doubles per wavefront
static_assert(kWaveLds >= 64 * kStageLd, "the MFMA staging tile must fit the wave's park region");
// Workgroup-cooperative D.X (throughput form): the 64 x 11 result rows of decision vector b are handed to
// wavefront b through slots PK_LV0 .. PK_FP7 of ITS OWN region ([node][11], stride 11 doubles: conflict-free row
// reads); slots PK_Q0 .. PK_DJJ are never touched by the hand-over, so a wavefront may park there while slower
// wavefronts of its workgroup still write their tiles.
constexpr int kCoopStageOff = PK_LV0 * 64;
static_assert(kWaveLds - kCoopStageOff >= 64 * 11, "the cooperative hand-over area must fit behind the early park slots");
constexpr int kXldsPipeFrom = kLongPhaseFrom;  // phases of this many nodes and more: state rows in double-buffered 44-row LDS slabs (below: one slab, staged at once)
// Residual-only instantiations park PK_Q0 .. PK_LV2 only: their region is the D.X operand image (cooperative LDS-staged
// form: a 68-row slab, or two 36-row vectors when two decision vectors share a wavefront), the hand-over area over it and
// then, once the wavefront has taken its rows, the park over that -- 6 KB instead of 9.5 KB per wavefront.
constexpr int kSlabRowsMax = 68, kPackRows = 36, kParkRes = (PK_LV2 + 1) * 64;
constexpr int kPipeSlabK = 8;    // k-steps per double-buffered slab of the long-phase form (32 state rows)
constexpr int wave_lds_doubles(bool jac, bool mfma, bool pack, bool split = false, bool longp = true) {
  return jac ? kWaveLds : (!mfma ? kParkRes : (split ? 64 * kStageLd /* one wavefront's own result tile */ : (pack ? kParkRes + 4 * 64 /* operand image (792), then park + the tile of the transposed residual stores */ : (longp ? 2 * 4 * kPipeSlabK * 11 + 64 * kPipeSlabK /* two images of a slab's state rows + the A ring */ : kSlabRowsMax * 11))));
}
static_assert(wave_lds_doubles(false, true, false, true) >= kParkRes, "residual-only split form: park over the result tile");
static_assert(wave_lds_doubles(false, true, true) >= 64 * 11 && wave_lds_doubles(false, true, false) >= 64 * 11 && wave_lds_doubles(false, true, false, false, false) >= 64 * 11, "hand-over area");
static_assert(wave_lds_doubles(false, true, true) >= kParkRes && wave_lds_doubles(false, true, false) >= kParkRes && wave_lds_doubles(false, true, false, false, false) >= kParkRes, "residual-only park");

// Compact Jacobian slots of a node (gel_host.hip walk_pattern() maps them to the reference's COO entries).  Only
// DISTINCT x-dependent values are stored: a tf column that is the exact negative of its t0 column, the node-uniform
// pos/velocity diagonal value (one scalar per phase, behind the phase's node slots) and the eight entries of the
// 4 x 4 quaternion block that do not depend on x (D[j][j+1] or 0: dq_c does not contain q_k) are restored by the
// consumer's gather map (gel_full_source) instead of being written.
//   0..2 pos/t (t0)   3..5 vel/mass   6..14 vel/position   [15..23 vel/velocity, air_fd]   s_vq +12 vel/quaternion
//   s_vt +6 vel/t (air_fd: t0, tf) or +3 (t0)   s_qq +8 quat/quaternion (k major: k < 2 -> rows 2,3; else rows 0,1)
//   s_qq+8 +8 quat/u   s_qq+16 +4 quat/t (t0)
constexpr int kSlotPT = 0, kSlotVM = 3, kSlotVP = 6;

#ifndef GEL_MIN_WAVES_PER_SIMD
#define GEL_MIN_WAVES_PER_SIMD 4  // 121-128 VGPRs: 4 waves/SIMD (16 per CU, matching the LDS budget); 5 spills heavily
#endif
#ifndef GEL_MIN_WAVES_PER_SIMD_PACKJAC
#define GEL_MIN_WAVES_PER_SIMD_PACKJAC 4  // two vectors per wavefront with derivatives (the per-half scalars are vector values there)
#endif
#ifndef GEL_MIN_WAVES_PER_SIMD_RES
#define GEL_MIN_WAVES_PER_SIMD_RES 5  // residual-only, two vectors per wavefront: 94 VGPRs, 25 KB of LDS per workgroup
                                      // (6: 80 VGPRs + 48 B of scratch per lane, 3 x 32 residual-only 29 % SLOWER, pooled A/B round 4)
#endif

typedef double gel_double4 __attribute__((ext_vector_type(4)));

#ifndef GEL_CA_CACHE
#define GEL_CA_CACHE 1  // the Mach interval of a node's first CA lookup serves its other aerodynamic-force evaluations
#endif
#ifndef GEL_XLDS_A_PF_JAC
#define GEL_XLDS_A_PF_JAC 4   // LDS-staged cooperative D.X, one vector per wavefront: A slabs in flight, fused launch.  A ring of four,
                              // the first four requested before the operand barrier.  In-process A/B at mixed-6x64 (tools/ab_inproc.py,
                              // round 5, 0.1 % spread): against all 17 of a 64-node phase at once (round 4's choice, made with a
                              // harness that could not see it) -1.2 %; 2 / 3 / 5 / 8 in flight +0.4 .. +0.6 % over 4, 6 level
#endif
#ifndef GEL_XLDS_A_PF_RES
#define GEL_XLDS_A_PF_RES 4  // ... residual-only launch (6x64: +11 % over 1; the fused launch does not care: +-1 % for 1..8)
#endif
#ifndef GEL_PACK_A_PRELOAD
#define GEL_PACK_A_PRELOAD 1  // two vectors per wavefront: all (<= 9) A slabs of D.X requested before the operand barrier
#endif


#ifndef GEL_XCD_RANGES
#define GEL_XCD_RANGES 1   // cooperative launches, work-item major order: every XCD takes contiguous runs of vector groups
#endif
#ifndef GEL_XCD_BLOCK
#define GEL_XCD_BLOCK 0    // groups per run (a power of two), 0: an eighth of the batch
#endif
#ifndef GEL_FRONT_BATCH
#define GEL_FRONT_BATCH 1  // cooperative forms: the kernel arguments of the walk to the phase record in one round trip, one chunk-record fetch
#endif
#ifndef GEL_DX_HOLD7
#define GEL_DX_HOLD7 1  // hold-type phases: D.X without the quaternion columns (two column tiles per wavefront instead of three)
#endif

#ifndef GEL_STORE_AUX
#define GEL_STORE_AUX 2  // cache policy of the Jacobian stores: 2 = nt (A/B: 0 plain, 1 sc0, 16 sc1, 18 sc1+nt)
#endif

#ifndef GEL_AERO_STORE_AUX
#define GEL_AERO_STORE_AUX 2  // cache policy of the fused kernel's aero-row stores: streamed (nt) like the Jacobian values.  In part A of the
                              // record a spec's row of a column is the phase's n nodes -- at n = 64 one store is eight whole 64-byte
                              // lines (in gel_eval_aero_all's layout the rows are n + 1 doubles long and every store of nodes 1 .. n
                              // began 8 bytes into a line and ended 8 bytes into another: the fused launch took 4.95 ms instead of
                              // 4.46).  In-process A/B at mixed-6x64: nt -0.8 % against ordinary stores.
#endif

// JAC: also the FD Jacobian.  MFMA: D.X on the matrix pipe (v_mfma_f64_16x16x4_f64) instead of VALU FMAs.
// SPLIT (latency form for a handful of decision vectors, e.g. the optimiser's B = 1 callback): every work item
// becomes four wavefronts -- part 0 does everything except the three position sweeps, parts 1..3 do the
// centre evaluation plus ONE position sweep each -- so the serial chain of a wavefront is about two trips of
// the atmosphere/geodesy chain instead of four plus the light sweeps.  Same expressions, same bits.
// PACK (cooperative form, every phase of the problem at most 32 nodes): a wavefront carries TWO decision vectors, one
// per 32-lane half, and a workgroup eight -- otherwise half of the lanes (and two of the four D.X row tiles) idle.
// The body is a device function of a (virtual) workgroup index: eval_kernel runs it for every workgroup of a launch,
// callback_kernel (gel_kernels.hip) for the first workgroups of the one launch that serves a whole callback.  In the forms
// that are not cooperative a workgroup may have any number of wavefronts (each its own work item).
// SPLITB (split form): k-steps of D.X whose state column is requested at once -- 17 (a whole 64-node phase) inside callback_kernel,
// which has registers to spare; 9 in the stand-alone launch, which is held to 128 VGPRs.
// AERO (throughput form with derivatives, one vector per wavefront; gel_eval_batch_aero_device): the lanes of an aerodynamic phase
// also write the aero path constraints of their state node (lib/con_aero.py:89-248,311-371) -- value and forward-difference
// gradient of the angle of attack, the dynamic pressure and their product -- from the centre evaluation and the position sweeps
// they run anyway: the geodetic -> atmosphere -> wind chain once per node instead of once here and once in aero_kernel.
template <bool JAC, bool MFMA, bool SPLIT = false, bool PACK = false, int SPLITB = 9, bool LONGP = true, bool NTS = true, bool AERO = false>
__device__ __forceinline__ void eval_body(const ProblemDev P, int B, const double* __restrict__ x, double* __restrict__ res,
                                          double* __restrict__ jvar, const unsigned vblk) {
  extern __shared__ double lds[];
  // the matrix-pipe forms are only launched when residual rows are asked for (eval_form(): mfma = use_mfma && want_res): knowing
  // that, the compiler drops the `if (rb)` tests of the cooperative forms and the thirty zero-initialisations in front of them
  if (MFMA) __builtin_assume(res != nullptr);
  // Cooperative forms: what the walk to the phase record needs of the kernel arguments, taken TOGETHER at the top -- the compiler
  // fetches an argument where it is first used, and the walk (workgroup -> work item -> chunk record -> phase record -> addresses
  // of the state rows) used them one or two at a time: three scalar-load round trips in a row before the chunk record was even
  // asked for, in front of every wavefront's first request to HBM.  One asm statement that wants them all in scalar registers
  // makes them one round trip (in-process A/B: -0.25 % mixed-6x64, -0.45 % 3 x 32 residual-only: such a round trip is a hit in the
  // scalar cache, 100-200 cycles; fetching the scalars of the LATER stages here as well costs registers and was not kept).
  int fB = B, f_chunk0 = P.chunk0, f_vmajor = P.vmajor, f_nchunks = P.nchunks;
  const int4* f_chunks = P.chunks;
  const PhaseDev* f_phases = P.phases;
  if (GEL_FRONT_BATCH && MFMA && !SPLIT) asm volatile("" : "+s"(fB), "+s"(f_chunk0), "+s"(f_vmajor), "+s"(f_nchunks), "+s"(f_chunks), "+s"(f_phases));
  const int park_off = P.park_off;
  // the cooperative forms meet at a barrier (operand image / hand-over) before any table lookup: no barrier of its own
  constexpr bool kSplitStage = MFMA && !SPLIT;   // cooperative forms: table entry requested now, written before their first barrier
  const double tab_mine = kSplitStage ? stage_tables_issue(P) : 0.0;
  const Tables tb = kSplitStage ? table_view(lds, P.Kw, P.Kc) : stage_tables(P, lds, true);
  const int lane = threadIdx.x & 63;
  // explicit LDS address space: ds_read/ds_write (lgkmcnt), never flat_* (which also ticks vmcnt)
  typedef __attribute__((address_space(3))) double lds_double;
  constexpr int kWL = wave_lds_doubles(JAC, MFMA, PACK, SPLIT, LONGP);   // this instantiation's region per wavefront
  constexpr int kHO = JAC ? kCoopStageOff : 0;             // where the cooperative hand-over area starts in it
  static_assert(!(MFMA && !SPLIT) || kWL - kHO >= 64 * 11, "the cooperative hand-over area must fit the region");
  lds_double* wave_lds = (lds_double*)lds + park_off + (threadIdx.x >> 6) * kWL;
  lds_double* park = wave_lds + lane;
  // Residual-only instantiations have a smaller region (the D.X operand image / hand-over area only, see
  // wave_lds_doubles()): once a wavefront has read its D.X rows from the hand-over area the region is private, and the
  // few slots such a launch parks (PK_Q0 .. PK_LV2) overlay it.
#define PARK_GET(slot) park[(slot) * 64]
#define PARK_SET(slot, val) park[(slot) * 64] = (val)

  // COOP: the throughput form with D.X on the matrix pipe.  A workgroup = ONE work item x FOUR decision vectors;
  // its wavefronts share the A operand (D) and form the product together (see phase A).
  constexpr bool COOP = MFMA && !SPLIT;
  static_assert(!AERO || (JAC && MFMA && !SPLIT && !PACK), "the aero rows ride in the cooperative form with derivatives, one vector per wavefront");
  static_assert(!COOP || kBlock == 256, "the cooperative D.X form is written for four wavefronts per workgroup");
  static_assert(!PACK || COOP, "two vectors per wavefront exist in the cooperative form only");
  constexpr int kVecWg = PACK ? 8 : 4;   // decision vectors per workgroup
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = PACK ? (lane >> 5) : 0;
  const int nb4 = (fB + kVecWg - 1) / kVecWg;  // COOP: workgroups per work item
  const long long item = COOP ? 0 : __builtin_amdgcn_readfirstlane((int)(((long long)vblk * blockDim.x + threadIdx.x) >> 6));
  if (!COOP && item >= (long long)B * (SPLIT ? P.nunits : P.nchunks)) return;
  // work-item major: all B vectors of one (phase, chunk) are neighbours, so the four wavefronts of a
  // workgroup cost the same (its LDS is only released when the slowest ends), and the list is ordered
  // dearest phase type first, so the tail of the launch drains with cheap wavefronts (measured -4 % on
  // the mixed vehicle at B = 4096, -6..9 % at B = 2048)
  // SPLIT: the list is walked in units = (work item, part), unit id = 4 * item + part, again with all B vectors
  // of a unit next to each other; P.unit0 / P.nunits select a range of units (unit-sharded launches)
  // COOP launch order: work-item major (all B vectors of one (phase, chunk) are neighbours, dearest phase type first), but
  // the chunks of ONE phase and ONE group of vectors are dispatched together and onto the same XCD (workgroup p goes to XCD
  // p % 8): every chunk of a phase reads all n + 1 state rows of its vectors and writes a block next to the other chunks'
  // blocks, so what the first one fetched is still in that XCD's L2 when the others ask, and the values of one vector
  // reach their DRAM pages together.  Measured on 12 x 128 (two chunks per phase), same box: +4.6 % with the chunks merely
  // adjacent in dispatch order; phases of at most 64 nodes have one chunk and keep plain work-item major order.
  int q, b0;
  int4 ck_it = int4{0, 0, 0, 0};
  bool ck_same = false;   // COOP: ck_it is this workgroup's record
  if (COOP) {
    const unsigned p = vblk, nb = (unsigned)nb4;
    unsigned it = p / nb, r = p - it * nb;
    if (f_vmajor) {
      // Vector-group major, XCD aware (meshes whose phases all have at most 32 nodes): the work items of ONE group of vectors run
      // as consecutive workgroups of ONE XCD (workgroup p lands on XCD p % 8), so the 128-byte lines that neighbouring phases'
      // state rows share in x -- and the partial lines of their residual rows -- meet in that XCD's L2 instead of being
      // fetched and written once per phase (3 x 32 residual-only: HBM traffic 1.11 -> 1.0x of the algorithmic bytes).
      const unsigned xcd = p & 7u, qq = p >> 3, items = (unsigned)f_nchunks;
      const unsigned gl = qq / items;
      it = qq - gl * items;
      r = gl * 8u + xcd;
      if (r >= nb) return;   // the last block of eight groups may be short (whole workgroup: before any barrier)
    }
    // the work item's WHOLE chunk record at once: with one chunk per phase (every phase of at most 64 nodes) it is the record this
    // workgroup works on -- no second fetch behind the first
    ck_it = load_const_int4(f_chunks + f_chunk0 + it);
    const int cw = __builtin_amdgcn_readfirstlane(ck_it.w);   // (position in the phase's run of chunks) << 16 | chunks
    const unsigned pos = (unsigned)cw >> 16, nc = (unsigned)cw & 0xffffu;
    unsigned c = pos, bg = r;
    if (nc > 1) {
      const unsigned l = pos * nb + r;                 // index inside the phase's nc * nb workgroups
      const unsigned blk = l / (8 * nc), rem = l - blk * 8 * nc;
      const unsigned m = min(8u, nb - blk * 8);        // vector groups in this block of eight (the last one may be short)
      c = rem / m;
      bg = blk * 8 + (rem - c * m);
    }
    // XCD-contiguous vector groups: workgroup p runs on XCD p % 8, so with the groups dealt in dispatch order every XCD touched
    // every eighth group of the batch -- every page of x / res / jvar that is being worked on, eight times over.  Within a block of
    // 8 G groups, XCD j instead takes the G consecutive groups j G .. (j + 1) G - 1 (for every work item alike: the same XCD comes
    // back to the same vectors' rows): each XCD's translations and lines cover an eighth of the pages.  G = GEL_XCD_BLOCK groups,
    // or a whole eighth of the batch (0: in-process A/B at B = 65536, mixed-6x64 -2.2 %, dense -4.6 %, 12 x 128 -1.0 %; runs of 64 / 512
    // groups -1.8 / -2.1 %, 0 / -0.7 %); the groups behind the last full block keep the dispatch order.  (Applied to the group index
    // the order ends with: the chunks of a long phase stay together on their XCD.)
    if (GEL_XCD_RANGES && !f_vmajor) {   // (the vector-group major order of the small-phase meshes: +1.0 % with it at 3 x 32, left alone)
      const unsigned G = GEL_XCD_BLOCK ? (unsigned)GEL_XCD_BLOCK : (nb >> 3), blk8 = 8u * G;
      if (G != 0) {
        const unsigned bq = bg / blk8, inb = bg - bq * blk8;   // blk8 a power of two when GEL_XCD_BLOCK is one: shifts
        if ((bq + 1u) * blk8 <= nb) bg = bq * blk8 + (p & 7u) * G + (inb >> 3);
      }
    }
    q = (int)(it - pos + c);
    ck_same = GEL_FRONT_BATCH && nc <= 1;
    b0 = (int)bg * kVecWg;                             // first vector of the workgroup
  } else {
    q = (int)(item / B);
    b0 = 0;
  }
  // COOP: a wavefront (PACK: a half) past the end of the batch (B not a multiple of 4 / 8) still computes its tiles for
  // the others; it reads vector B - 1 and leaves after the hand-over without writing anything
  const int bw = PACK ? b0 + 2 * wv + half : b0 + wv;
  const bool ghost = COOP && bw >= fB;
  const bool pack_full = PACK && __builtin_amdgcn_ballot_w64(ghost) == 0;   // both vectors of this wavefront exist (wave-uniform)
  const int b = COOP ? min(bw, fB - 1) : (int)(item - (long long)q * B);
  const int ci = SPLIT ? ((P.unit0 + q) >> 2) : q;
  const int part = SPLIT ? ((P.unit0 + q) & 3) : 0;  // wave-uniform
  int4 ck = ck_it;
  if (!ck_same) ck = COOP ? load_const_int4(f_chunks + f_chunk0 + ci) : P.chunks[(SPLIT ? 0 : P.chunk0) + ci];   // wave-uniform
  const int sec = __builtin_amdgcn_readfirstlane(ck.x);
  const int j0 = __builtin_amdgcn_readfirstlane(ck.y);
  const int dsw = __builtin_amdgcn_readfirstlane(ck.z);  // first gel_double4 of this work item in Dsw
  const int j = PACK ? (lane & 31) : j0 + lane;  // node inside the phase (PACK: one chunk per phase, j0 = 0)
  const PhaseDev ph = load_phase((COOP ? f_phases : P.phases) + sec);  // by value, in SGPRs, before any store
  // Latency form of a WHOLE one-vector VALUES-ONLY evaluation (the optimiser's objfunc): the four wavefronts of a workgroup are the
  // four parts of ONE work item, of which only the lead has anything to do -- so each forms one 16-row tile of D.X (17 instead of 68
  // matrix instructions in the lead; the state rows cross the bus once, a quarter per wavefront); the tiles meet in the lead's
  // staging tile at a workgroup barrier and the other parts leave.  6 x 64: values-only callback 24.0 -> 20.4 us, objfunc through
  // the Python mirrors 36.8 -> 33.3 us.  With derivatives the launch is bounded by the PCIe drain of its 240 KB of results, and the
  // barriers cost small phases more than the split saves (example: 26.2 -> 29.3 us): the lead multiplies alone there.
  const bool csplit = SPLIT && MFMA && !JAC && B == 1 && res != nullptr && P.split_vel && !((P.unit0 | P.nunits) & 3);   // wave-uniform (kernel arguments)
  const bool only_dx = SPLIT && part && (!ph.air || !JAC);
  if (only_dx && !csplit) return;  // only aerodynamic phases have the long position sweeps (and only with derivatives)
  const bool lead = !(SPLIT && part);     // the wavefront that owns everything but the split-off sweeps
  const int n = ph.n;
  // the matrix pipe reads all 64 lanes: lanes past the end of a ragged phase stay alive (with zero
  // operands and clamped addresses) until the D.X product is done
  const bool active = j < n;
  if (!MFMA && !active) return;
  const int jc = active ? j : n - 1;  // clamped node for address formation
  const int g = ph.ua + j;       // global collocation node
  const int xj = ph.xa + 1 + jc;  // its state row (x-node j+1)
  const int M = P.M, N = P.N;

  const double* xb = x + (size_t)b * P.nvars;
  const double* xm = xb;
  const double* xr = xb + M;
  const double* xv = xb + 4 * M;
  const double* xq = xb + 7 * M;
  const double* xu = xb + 11 * M;
  const double* xt = xb + 11 * M + 2 * N;
  // wave-uniform scalars live in SGPRs (two decision vectors per wavefront: per half, so they stay vector values there)
#define GEL_UNI(v) (PACK ? (v) : wave_uniform(v))
  // The phase's two knot times: requested here, TAKEN (v_readfirstlane: the first wait of the wavefront) only once every other
  // load of phase A has been requested -- one HBM round trip instead of two in a row in front of the D.X product.
#ifndef GEL_KNOT_VLOAD
#define GEL_KNOT_VLOAD 1   // the knot times by a VECTOR load (the index made opaque): a wave-uniform address becomes a scalar load, which the
                           // compiler -- short of scalar registers here -- waits for on the spot (to park the pair in a VGPR's lanes): an HBM
                           // round trip IN FRONT of the requests for the state rows, the very thing the comment above is about
#endif
  int sec_ld = sec;
  // (the residual-only forms have the scalar registers to let the pair wait with the others; the latency form reads x from pinned
  // host memory, where the pair as a vector load cost a one-vector evaluation with derivatives 2.5 us: 31.6-34.3 -> 36.2-37.1)
  if (GEL_KNOT_VLOAD && JAC && !SPLIT) asm volatile("" : "+v"(sec_ld));
  const double to_ld = xt[sec_ld], tf_ld = xt[sec_ld + 1];
  double to = 0.0, tf = 0.0, fds = 0.0, fdt = 0.0;
  const double dx = P.dx, ut = P.ut;
  // unit_t / 2 (exact): v * unit_t / 2.0 as the reference writes it is round(v unit_t) / 2 = round(v (unit_t / 2)) -- scaling by a power
  // of two commutes with the rounding -- so `v * hT` has the same bits with one multiplication less per value
  const double hT = P.hT;
  const double inv_dx = P.inv_dx;
  // Jacobian entry from a perturbed/centre pair: -(f_p - f_c)/dx*(tf-to)*unit_t/2  (con_dynamics.py:372),
  // as (f_c - f_p) times the wave-uniform scale (tf-to)*unit_t/2/dx
#define GEL_TAKE_KNOT_TIMES()                                   \
  do {                                                          \
    to = GEL_UNI(to_ld); tf = GEL_UNI(tf_ld);                   \
    fds = GEL_UNI(inv_dx * (tf - to) * ut / 2.0);               \
    fdt = GEL_UNI(inv_dx * ut / 2.0);                           \
  } while (0)
  // NaN / Inf detector: lanes that wrote a non-finite value, accumulated on the scalar unit (a running per-lane sum would hold
  // two VGPRs for the whole kernel).  What is tested: every residual row a wavefront writes -- D is dense, so a non-finite state
  // column reaches every row of its phase, and mass, thrust, force and gravity reach the velocity rows through f_c -- and, in a
  // wavefront that writes no residual rows (Jacobian-only calls, the split-off position sweeps of the latency form), every Jacobian
  // value.  The fused launch used to test all 63 stored values of a node as well: 49 compares + 49 scalar ORs per wavefront for
  // values that are differences and products of numbers the residual rows already vouch for (the reference tests nothing:
  // Trajectory_Optimization.py:240,311 hard-code fail = False).
  unsigned long long bad = 0;
#define GEL_CHK(v) (bad |= __builtin_amdgcn_ballot_w64(!(fabs(v) <= 1.79769313486231570815e308)))
  // PACKED (split form only, P.shard_width != 0: gel_eval_shard_packed_device): every unit writes ITS entries of a decision
  // vector as one contiguous block -- compact Jacobian slots first ([slot][node of the chunk], only the slots the unit owns),
  // then the phase scalar (first chunk of a phase), then the residual rows mass | position | velocity | quaternion of the
  // chunk's nodes -- at unit_base[unit] inside the rank's block of the vector, `res` being the rank's slice [B][shard_width]
  // of the exchange buffer (gel_shard_plan has the map back to the reference layouts).
  const bool packed = SPLIT && P.shard_width != 0;
  const int nn = min(64, n - j0);                         // nodes of this chunk
  // slots a packed unit skips in front of the ones it owns: part k > 0 owns the three slots of position column k - 1; part 0
  // of an aerodynamic phase everything but the nine position-sweep slots 6 .. 14
  const int sub_hi = packed ? (part ? kSlotVP + 3 * (part - 1) : (ph.air ? 9 : 0)) : 0;
  const int sub_lo = (packed && part) ? sub_hi : 0;
  const int pk_nj = (ph.K - sub_hi) * nn + (j0 == 0 ? 1 : 0);   // part 0: doubles of its Jacobian part (residual rows follow)
  double* const pk_out = packed ? res + (size_t)b * P.shard_width + load_const(P.unit_base + P.unit0 + q) : nullptr;
  double* rb = packed ? (lead ? pk_out : nullptr) : ((res && lead) ? res + (size_t)b * 11 * N : nullptr);
  const int gn = packed ? lane : g;                       // node index inside a residual group
  const int rs_m = packed ? pk_nj : 0, rs_p = packed ? pk_nj + nn : N, rs_v = packed ? pk_nj + 4 * nn : 4 * N,
            rs_q = packed ? pk_nj + 7 * nn : 7 * N;
  // Buffer store: wave-uniform resource (base = this wavefront's first value), lane offset in a VGPR, slot offset
  // on the scalar unit -- no vector address arithmetic at all (69 v_lshl_add_u64 gone, 4 VGPRs less).  Streamed
  // once and never re-read by this kernel: non-temporal.  Measured: nt over plain stores -2..4 % in round 1; with the
  // round-2 kernel (less arithmetic per stored byte) -7 % on mixed-6x64 and -15 % on 12x128 (sc0 like plain, sc1 and
  // sc1+nt +25..45 % slower); buffer form over global_store another -1.8 %.
  typedef unsigned gel_u2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t jrs =
      // a phase's node values are laid out [64-node chunk][slot][node of the chunk]: this wavefront's block starts at j0 * K
      __builtin_amdgcn_make_buffer_rsrc(JAC ? (packed ? (void*)pk_out : (void*)(jvar + (size_t)(PACK ? min(b0 + 2 * wv, B - 1) : b) * P.V + ph.voff + (size_t)j0 * ph.K)) : (void*)nullptr,
                                        0, -1, 0x00020000);
  const int jvo = PACK ? (half * (int)P.V + j) * 8 : lane * 8;
  const int cw8 = nn * 8;   // bytes between two slots of this wavefront's block
#define EMIT_AT(byteoff, val)                                                           \
  do {                                                                                  \
    const double _v = (val);                                                            \
    gel_u2 _d;                                                                          \
    __builtin_memcpy(&_d, &_v, 8);                                                      \
    __builtin_amdgcn_raw_buffer_store_b64(_d, jrs, jvo, (byteoff), NTS ? GEL_STORE_AUX : 0); \
    if (!rb) GEL_CHK(_v);                                                                 \
  } while (0)
#define EMIT(slot, val) EMIT_AT(((int)(slot) - (SPLIT ? (((int)(slot) >= kSlotVP) ? sub_hi : sub_lo) : 0)) * cw8, val)
  // AERO: this phase's record (scalar loads, once per group of entries: GEL_AERO_FETCH), the vector's output record as a buffer
  // resource (wave-uniform base; lane offset jvo; block offsets on the scalar unit), streamed like the Jacobian values.  A store of
  // node j0 + lane + 1, column c of a block whose first column is c0: base[kind] + ((c0 + c) n + j0) 8 bytes into the record
  // (part A, spec-major: gel_device.h AeroPhaseDev).
  const AeroPhaseDev* const aph = AERO ? P.aero_ph + sec : nullptr;
  const int akinds = AERO ? __builtin_amdgcn_readfirstlane(load_const(&aph->kinds)) : 0;   // 0: no aero rows in this phase
  const __amdgpu_buffer_rsrc_t ars =
      __builtin_amdgcn_make_buffer_rsrc(AERO ? (void*)(P.aero_out + (size_t)b * P.aero_ld) : (void*)nullptr, 0, -1, 0x00020000);
#define AEMIT(byteoff, val)                                                              \
  do {                                                                                  \
    const double _v = (val);                                                            \
    gel_u2 _d;                                                                          \
    __builtin_memcpy(&_d, &_v, 8);                                                      \
    __builtin_amdgcn_raw_buffer_store_b64(_d, ars, jvo, (byteoff), GEL_AERO_STORE_AUX); \
  } while (0)
// (latency form: the rows go to pinned host memory -- streamed, so that they cross PCIe while the wavefront computes on instead of
// waiting in L2 for the end of the kernel; throughput form: the strided rows' partial lines want to meet in L2 first)
#define RSTORE(idx, val) do { if (SPLIT) __builtin_nontemporal_store((val), rb + (idx)); else rb[idx] = (val); } while (0)
  // Residual rows as CONTIGUOUS stores.  The reference's layout is node-major ([node][x y z], [node][w x y z]): written lane =
  // node, a store instruction puts 8 bytes every 24 / 32 bytes -- 24 / 32 partial 64-byte write requests per instruction, three or
  // four instructions per line.  A full 64-node chunk instead turns its w values per node through an LDS tile (node-major in,
  // 64-wide rows out: one wavefront's LDS operations execute in order, no barrier) and writes w contiguous 512-byte segments:
  // L1 -> L2 write requests per evaluation 3,367 -> 3,047 at mixed-6x64 (TCP_TCC_WRITE_REQ).  What it buys in TIME is inside the noise of where the
  // driver places a process's 14 GB of buffers (+-3 % between processes of one build; pooled A/B of round 4: +-1 % against the
  // strided form) -- the launch is bound by package power, and the bytes are the same.  Measured on the way (same box, round 4):
  // the residual rows are 17 % of the stored bytes and, removed entirely, 9-10 % of the launch (3.41 -> 3.06 ms; without any store
  // 2.43 ms).  Ragged chunks (lanes past the phase have left) keep the strided form.  -DGEL_RES_XPOSE=0: strided everywhere (A/B).
#ifndef GEL_RES_XPOSE
#define GEL_RES_XPOSE 1
#endif
  // Two vectors per wavefront (PACK): the tile holds half 0's rows then half 1's ([w * lane + c] does that by itself); row i of
  // the tile belongs to the vector of half (64 i + lane) / (32 w), so a lane may store the OTHER half's values: only in a
  // wavefront whose two vectors both exist (no ghost half) and whose phase has exactly 32 nodes.
  constexpr int kTileOff = JAC ? PK_FP0 * 64 : kParkRes;     // phase A: slots FP0.. are not in use yet (residual-only: behind the park)
  constexpr bool kCanXpose = (GEL_RES_XPOSE != 0) && (kWL >= kTileOff + 4 * 64);
  // whole lines, written once and never read by this kernel: non-temporal like the Jacobian values (the strided form's partial
  // lines want to meet in L2 first: written non-temporally they cost 4 % more HBM writes, round 2)
#define RSTORE_LINE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define xpose (kCanXpose && (PACK ? (cw8 == 256 && pack_full) : cw8 == 512))   /* wave-uniform: every lane holds a node (cw8 = 8 nn is in an SGPR anyway) */
#define RSTORE_ROWS(tile, w, rs_off, vals)                                                                   \
  do {                                                                                                       \
    _Pragma("unroll") for (int c_ = 0; c_ < (w); c_++) GEL_CHK((vals)[c_]);                                  \
    if (xpose) {                                                                                             \
      lds_double* t_ = (tile);                                                                               \
      _Pragma("unroll") for (int c_ = 0; c_ < (w); c_++) t_[(w) * lane + c_] = (vals)[c_];                   \
      _Pragma("unroll") for (int i_ = 0; i_ < (w); i_++) {                                                   \
        const double v_ = t_[64 * i_ + lane];                                                                \
        if (PACK) {                                                                                          \
          const int e_ = 64 * i_ + lane, h_ = (e_ >= 32 * (w)) ? 1 : 0;                                      \
          double* rbh_ = res + (size_t)(b0 + 2 * wv + h_) * 11 * N;                                          \
          RSTORE_LINE(rbh_ + (rs_off) + (w) * ph.ua + (e_ - 32 * (w) * h_), v_);                             \
        } else {                                                                                             \
          RSTORE_LINE(rb + (rs_off) + (w) * (gn - lane) + 64 * i_ + lane, v_);                               \
        }                                                                                                    \
      }                                                                                                      \
    } else {                                                                                                 \
      _Pragma("unroll") for (int c_ = 0; c_ < (w); c_++) RSTORE((rs_off) + (w) * gn + c_, (vals)[c_]);       \
    }                                                                                                        \
  } while (0)
  const double inv_uv = P.inv_uv;   // 1 / unit_v, divided on the host
#define FDQ(fp, fc) (((fc) - (fp)) * fds)
  // COO-DIRECT output (latency form, the optimiser's B = 1 callback: gel_host.hip run_host / gel_eval_callback).  The blocks of the
  // reference's COO value vector whose entries are ALL x-dependent -- pos/velocity, pos/t, vel/mass, vel/position, vel/quaternion,
  // vel/t, quat/u, quat/t (lib/con_dynamics.py:180-195,373-400,431-449,482-489,600-625: runs of [node][xyz] or [node][wxyz]) -- are
  // written straight to their places in a FULL value array in pinned host memory (P.coo_full; the constants lie there already),
  // each group of W slots of a node as W coalesced stores through an LDS tile, a negated twin run (the tf column of a t0 column)
  // W n entries behind.  What is left for the host to scatter are the entries that sit alone between constants (the diagonal of the
  // dense velocity block, the pairs of the dense quaternion block): 11 slots per node instead of 49.  P.coo[8 * phase + g]: first
  // entry of group g's run in the full array (g: 0 pos/velocity, 1 pos/t, 2 vel/mass, 3 vel/position (+ 3 n k), 4 vel/quaternion
  // (+ 3 n k), 5 vel/t, 6 quat/u (+ 4 n k), 7 quat/t).
  const bool coo = SPLIT && !packed && P.coo_full != nullptr;   // wave-uniform
  enum { CG_PV = 0, CG_PT, CG_VM, CG_VP, CG_VQ, CG_VT, CG_QU, CG_QT };
#define GEL_COO_BASE(g) ((size_t)load_const(P.coo + 8 * sec + (g)))
#define GEL_COO_EMIT(W, base_, vals_, okl_, twin_)                                                                     \
  do {                                                                                                                 \
    double* d_ = P.coo_full + (base_) + (size_t)(W) * j0;                                                              \
    const unsigned long long okm_ = __builtin_amdgcn_ballot_w64(okl_);                                                 \
    _Pragma("unroll") for (int c_ = 0; c_ < (W); c_++) if (!rb) GEL_CHK((vals_)[c_]);                                  \
    if (cw8 == 512) {   /* a full chunk: every lane holds a node, the tile turns [node][W] into W rows of 64 */      \
      lds_double* t_ = wave_lds + kTileOff;                                                                            \
      _Pragma("unroll") for (int c_ = 0; c_ < (W); c_++) t_[(W) * lane + c_] = (vals_)[c_];                            \
      _Pragma("unroll") for (int i_ = 0; i_ < (W); i_++) {                                                             \
        const int e_ = 64 * i_ + lane;                                                                                 \
        const double v_ = t_[e_];                                                                                      \
        const int ln_ = ((W) == 4) ? (e_ >> 2) : ((e_ * 171) >> 9);   /* the lane (node) that owns entry e_ */          \
        if ((okm_ >> ln_) & 1ull) {   /* streamed: on its way over PCIe while the wavefront computes on */                \
          __builtin_nontemporal_store(v_, d_ + e_);                                                                    \
          if (twin_) __builtin_nontemporal_store(-v_, d_ + e_ + (W) * n);                                              \
        }                                                                                                              \
      }                                                                                                                \
    } else if (okl_) {   /* ragged chunk (lanes past the phase have left): every lane writes its own W entries */      \
      _Pragma("unroll") for (int c_ = 0; c_ < (W); c_++) {                                                             \
        d_[(W) * lane + c_] = (vals_)[c_];                                                                             \
        if (twin_) d_[(W) * (n + lane) + c_] = -(vals_)[c_];                                                           \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
  // W consecutive compact slots of this node that are one [node][W] run of a COO block (group g, k-th run of the group)
#define EMIT_GROUP(W, slot0, vals_, g, k, twin_, okl_)                                                                 \
  do {                                                                                                                 \
    if (coo) {                                                                                                         \
      GEL_COO_EMIT(W, GEL_COO_BASE(g) + (size_t)(W) * n * (k), vals_, okl_, twin_);                                    \
    } else if (okl_) {                                                                                                 \
      _Pragma("unroll") for (int c_ = 0; c_ < (W); c_++) EMIT((slot0) + c_, (vals_)[c_]);                              \
    }                                                                                                                  \
  } while (0)

  // ======================= phase A: every global load =======================
  // XLDS (cooperative form, phases whose n + 1 state rows fit one 68-row slab): the four decision vectors' state rows are
  // staged in LDS once per workgroup and serve both as the B operand of D.X and as the node's own state -- no strided
  // 8-byte global loads in the matrix loop.  Measured against fetching B per k-step from global memory (same box,
  // B = 16384 / 65536): fused launch -2 % at 6x64, residual-only -9 % at 3x32; with two slabs (n = 128) the extra
  // workgroup barriers cost what the loads save (residual-only +10 %), so longer phases keep the global form.
  constexpr bool XLDS = COOP;   // the cooperative forms stage the state rows in LDS: the node's own state row comes from there
  double me = 0.0, re[3] = {0.0, 0.0, 0.0};
  if (!XLDS) { me = xm[xj]; re[0] = xr[3 * xj]; re[1] = xr[3 * xj + 1]; re[2] = xr[3 * xj + 2]; }
  const double tau = P.tau[ph.toff + jc];
  {
    double q[4] = {0.0, 0.0, 0.0, 0.0}, ve[3] = {0.0, 0.0, 0.0};
    if (!XLDS) {
#pragma unroll
      for (int c = 0; c < 4; c++) q[c] = xq[4 * xj + c];
#pragma unroll
      for (int c = 0; c < 3; c++) ve[c] = xv[3 * xj + c];
    }
    double u0 = 0.0, u1 = 0.0;
    if (!ph.hold) { u0 = xu[2 * (ph.ua + jc)]; u1 = xu[2 * (ph.ua + jc) + 1]; }
    const double djj = JAC ? P.Dt[ph.doff + (size_t)(jc + 1) * n + jc] : 0.0;  // D[j][j+1]
    // The LAST state row of a phase on the vector unit.  A phase has n + 1 state rows; with n a multiple of four (every mesh of
    // BASELINE.json, the usual case) the matrix pipe's last k-step would multiply ONE real row and three rows of zeros: three
    // v_mfma_f64_16x16x4_f64 (192 cycles of the fp64 datapath) for what twelve v_fma_f64 (48 cycles) do on the accumulators as
    // they lie (cooperative forms: C[row (l>>4) + 4i][col l&15] += D[row][n] X[n][col] -- four loads from D's last column, three
    // broadcast reads of the row; latency form: after the transpose, lane = node, eleven fused operations).  Every matrix-pipe form
    // adds rows 0 .. n - 1 by k-steps in ascending order and then row n by one fused operation: bit-identical to one another.
    const bool tail1 = MFMA && ((n + 1) & 3) == 1 && n >= 4;   // wave-uniform
    const int ksteps = tail1 ? (n >> 2) : ((n + 4) >> 2);      // k-steps of the matrix pipe: ceil((n+1)/4), or n/4 + the tail row
    const double* const dcol = P.Dt + ph.doff + (size_t)n * n; // D[.][n]
    double dl4[4] = {0, 0, 0, 0};                               // cooperative forms: D[row][n] of this lane's four accumulator rows
    if (COOP && tail1) {
      const int rt = PACK ? (wv & 1) : wv;
#pragma unroll
      for (int i = 0; i < 4; i++) dl4[i] = dcol[min(j0 + 16 * rt + (lane >> 4) + 4 * i, n - 1)];   // j0: this work item's chunk of the phase
    }
    const double dlast = (SPLIT && tail1) ? dcol[jc] : 0.0;     // latency form: D[j][n]
    double xl[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // latency form: state row n
#define GEL_DX_TAIL_ACC(nct_, xn_of_ct)                                                             \
  do {                                                                                              \
    if (tail1) {                                                                                    \
      _Pragma("unroll") for (int ct = 0; ct < 3; ct++) {                                            \
        if (ct < (nct_)) {   /* wave-uniform: column tiles in use */                                \
          const double xn_ = (xn_of_ct);                                                            \
          _Pragma("unroll") for (int i = 0; i < 4; i++) acc[ct][i] = __builtin_fma(dl4[i], xn_, acc[ct][i]); \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
  } while (0)

    // the reference rows of engine-off / hold phases (state row 0 of the phase): from the LDS image of the state rows where a
    // form stages one (no load at all), else fetched after the product
    double m0 = 0.0, q0[4] = {0, 0, 0, 0};
    bool ref0_done = false;
    // D.X rows (lib/con_dynamics.py:54,146,256,524)
    double lm = 0.0, lr[3] = {0, 0, 0}, lv[3] = {0, 0, 0}, lq[4] = {0, 0, 0, 0};
    if (rb || csplit) {
      if (XLDS && (PACK || !LONGP || n < kXldsPipeFrom)) {   // !LONGP: the launcher vouches that no phase is longer
        // [64 x (n+1)] . [(n+1) x 44] per WORKGROUP, operands as in the branch below, but B comes from LDS: every wavefront
        // stages ITS OWN vector's n + 1 <= 68 state rows (17 k-steps) -- lane = row, the eleven interleaved columns (mass |
        // pos xyz | vel xyz | quat wxyz) side by side, [row][11] at the start of its region -- and all four read the 44
        // packed columns from there.  Eleven strided global loads per wavefront instead of three per k-step, and the node's
        // own state row is read back from the same image.
        const int c16 = lane & 15, kq = lane >> 4;
        constexpr int kSlabK = 17, kSlabRows = 4 * kSlabK;
        static_assert(kSlabRows == kSlabRowsMax, "slab size");
        static_assert(!COOP || kSlabRows * 11 <= kWL, "a slab of state rows must fit the wave's park region");
        lds_double* regions = (lds_double*)lds + park_off;
        // PACK: [32 x (n+1)] . [(n+1) x 88]: row tiles 0 and 1 only, six column tiles; wavefront w forms row tile w & 1 for
        // column tiles 3 (w >> 1) .. + 2; vector v's image lies in wavefront v >> 1's region, half v & 1.
        // Hold-type phases (attitude hold / vertical: quaternion rows q[1:] - q[0], lib/con_dynamics.py:521-522) never read the
        // quaternion columns of D.X: their vectors are packed seven columns each (mass | position | velocity) -- 28 columns in TWO
        // column tiles per wavefront instead of 44 in three, a third of the phase's matrix instructions less (one vector per wavefront only).
        // A column's sums do not depend on its neighbours in the tile: the columns that are formed keep their bits.
        const bool h7 = GEL_DX_HOLD7 && !PACK && ph.hold;   // wave-uniform (one work item per workgroup); PACK: not taken (3 x 32 residual-only +1.1 % with it)
        const int ncv = h7 ? 7 : 11;               // columns per vector
        const int ncols = ncv * kVecWg;
        const int ct0 = PACK ? (h7 ? 2 : 3) * (wv >> 1) : 0;
        int xoff[3];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
          const int c = 16 * (ct0 + ct) + c16;    // packed column: vector c / ncv, state column c % ncv
          const int vb = min(h7 ? c / 7 : c / 11, kVecWg - 1); // the last columns are padding: computed on the last vector, never read
          const int col = (c < ncols) ? c - ncv * vb : 0;
          xoff[ct] = PACK ? (vb >> 1) * kWL + (vb & 1) * kPackRows * 11 + col + kq * 11 : vb * kWL + col + kq * 11;
        }
        gel_double4 acc[3];
        const double* ap = P.Dst + (size_t)dsw * 4 + (PACK ? (wv & 1) : wv) * 64 + lane;
        // ksteps <= kSlabK: the phase is one slab
        // Every load of phase A is REQUESTED before anything waits: the A slabs first (they do not depend on x: L2), then the
        // state rows (HBM), and only then the first s_waitcnt of the wavefront.  A slabs: all k-steps of the phase at once
        // (kAAll), or a ring of kAPF in flight.
        constexpr int kAPF = JAC ? GEL_XLDS_A_PF_JAC : GEL_XLDS_A_PF_RES;
        constexpr int kAllN = PACK ? 9 : kSlabK;
        constexpr bool kAAll = PACK ? (GEL_PACK_A_PRELOAD != 0) : (kAPF >= kSlabK);
        double a_all[kAAll ? kAllN : 1], a_ring[kAAll ? 1 : kAPF];
        if (kAAll) {
#pragma unroll
          for (int ks = 0; ks < kAllN; ks++) a_all[ks] = ap[min(ks, ksteps - 1) * 256];
        } else {
#pragma unroll
          for (int i = 0; i < kAPF; i++) a_ring[i] = ap[min(i, ksteps - 1) * 256];
        }
        // State rows: lane = row for rows 0 .. 63 (eleven loads, addresses clamped into the phase: no branch around a load);
        // the few rows behind them (64 .. 67; two vectors per wavefront: 64 .. 71) element by element (one or two loads).  Rows past the phase meet zero columns of D.
        constexpr int kRowsStaged = PACK ? 2 * kPackRows : kSlabRows, kExtra = (kRowsStaged - 64) * 11;
        double st[11], sx[(kExtra + 63) / 64];
        {
          const int hv = PACK ? lane / kPackRows : 0;          // PACK: which of the wavefront's two vectors
          const int k = PACK ? lane - kPackRows * hv : lane;
          const double* xs = PACK ? x + (size_t)min(b0 + 2 * wv + hv, B - 1) * P.nvars : xb;
          const int xk = ph.xa + min(k, n);
          st[0] = xs[xk];
#pragma unroll
          for (int c = 0; c < 3; c++) { st[1 + c] = xs[M + 3 * xk + c]; st[4 + c] = xs[4 * M + 3 * xk + c]; }
#pragma unroll
          for (int c = 0; c < 4; c++) st[7 + c] = xs[7 * M + 4 * xk + c];
          // the rows behind them: element e = lane + 64 i is row 64 + (e mod kER) (kER = 4 or 8 rows: a power of two), column e / kER --
          // shifts and masks instead of divisions by 11 and by the rows of a half; with two vectors per wavefront these rows all
          // belong to the second vector (rows 36 .. 71 of the image)
          constexpr int kER = kRowsStaged - 64, kERs = (kER == 8) ? 3 : 2;
          static_assert(kER == (1 << kERs) && (!PACK || 64 >= kPackRows), "extra rows: a power of two, all in the second half");
#pragma unroll
          for (int i = 0; i < (kExtra + 63) / 64; i++) {
            const int e = min(lane + 64 * i, kExtra - 1);
            const int rr = 64 + (e & (kER - 1)), c = e >> kERs;
            const int kx = min(PACK ? rr - kPackRows : rr, n);
            const double* xe = PACK ? x + (size_t)min(b0 + 2 * wv + 1, B - 1) * P.nvars : xb;
            const int xr_ = ph.xa + kx;
            const int off = (c == 0) ? xr_ : ((c < 4) ? M + 3 * xr_ + (c - 1) : ((c < 7) ? 4 * M + 3 * xr_ + (c - 4) : 7 * M + 4 * xr_ + (c - 7)));
            sx[i] = xe[off];
          }
          // ---- first wait of the wavefront ----
          // A row past the phase (k > n) was fetched from the clamped address, i.e. it holds a copy of row n: it meets a zero
          // column of D, and 0 * (a finite value) adds nothing -- as in the split form, which clamps the same way.  (If row n is not
          // finite the product is not either, with or without the copies: row n has non-zero entries of D in every node's row.)
          lds_double* dst = wave_lds + lane * 11;
#pragma unroll
          for (int c = 0; c < 11; c++) dst[c] = st[c];
#pragma unroll
          for (int i = 0; i < (kExtra + 63) / 64; i++) {
            // no branch: the lanes past the last element were given its address (clamped above) and write its value once more -- a
            // lane-conditional store drew its load into the branch, behind a wait of its own (two vectors per wavefront: an extra
            // round trip in front of the operand barrier)
            const int e = min(lane + 64 * i, kExtra - 1);
            wave_lds[(64 + (e & (kER - 1))) * 11 + (e >> kERs)] = sx[i];
          }
        }
        stage_tables_commit(P, lds, tab_mine);
        __syncthreads();
        const int klast = ksteps;
        // (the loops exist twice, with and without the third column tile: one scalar branch per phase instead of one per k-step)
#define GEL_DX_ONE_SLAB(kTiles)                                                                                          \
  do {                                                                                                                   \
    if (kAAll) {                                                                                                         \
      _Pragma("unroll") for (int ks = 0; ks < kAllN; ks++) {                                                             \
        if (ks == 0 || ks < klast) {   /* wave-uniform; a phase has at least one k-step (the no-D.X ablation multiplies one, too) */ \
          const int ro = ks * 44;      /* 4 rows of 11 columns per k-step */                                             \
          /* the first k-step takes the constant 0 as its C operand (an inline operand of the instruction) instead of    \
             accumulators that twenty-four v_mov have cleared */                                                         \
          const gel_double4 zero4 = gel_double4{0.0, 0.0, 0.0, 0.0};                                                     \
          double bl_[kTiles];                                                                                            \
          _Pragma("unroll") for (int ct = 0; ct < (kTiles); ct++) bl_[ct] = regions[xoff[ct] + ro];                      \
          _Pragma("unroll") for (int ct = 0; ct < (kTiles); ct++)                                                        \
            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_all[ks], bl_[ct], ks ? acc[ct] : zero4, 0, 0, 0);           \
        }                                                                                                                \
      }                                                                                                                  \
    } else {                                                                                                             \
      /* k-steps of A slabs in flight ahead of the matrix pipe (an L2 round trip each; the first ones requested before the barrier) */ \
      _Pragma("unroll") for (int ct = 0; ct < 3; ct++) acc[ct] = gel_double4{0.0, 0.0, 0.0, 0.0};                        \
      for (int ks = 0; ks < klast; ks += kAPF) {                                                                         \
        _Pragma("unroll") for (int i = 0; i < kAPF; i++) {                                                               \
          if (ks + i < klast) {   /* wave-uniform */                                                                     \
            const int ro = (ks + i) * 44;                                                                                \
            double bl_[kTiles];                                                                                          \
            _Pragma("unroll") for (int ct = 0; ct < (kTiles); ct++) bl_[ct] = regions[xoff[ct] + ro];                    \
            _Pragma("unroll") for (int ct = 0; ct < (kTiles); ct++)                                                      \
              acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_ring[i], bl_[ct], acc[ct], 0, 0, 0);                      \
          }                                                                                                              \
          a_ring[i] = ap[min(ks + i + kAPF, ksteps - 1) * 256];                                                          \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
  } while (0)
        if (h7) GEL_DX_ONE_SLAB(2); else GEL_DX_ONE_SLAB(3);
#undef GEL_DX_ONE_SLAB
        GEL_DX_TAIL_ACC(h7 ? 2 : 3, regions[xoff[ct] + (n - kq) * 11]);   // row n of this lane's column (xoff points at row kq)
        {
          // the node's own state row comes from the same image (after the product: nothing of it is live across the loop)
          lds_double* src = wave_lds + (half * kPackRows + jc + 1) * 11;
          me = src[0];
#pragma unroll
          for (int c = 0; c < 3; c++) { re[c] = src[1 + c]; ve[c] = src[4 + c]; }
#pragma unroll
          for (int c = 0; c < 4; c++) q[c] = src[7 + c];
          lds_double* src0 = wave_lds + (half * kPackRows) * 11;   // state row 0 of the phase
          if (!ph.engine_on) m0 = src0[0];
          if (ph.hold) {
#pragma unroll
            for (int c = 0; c < 4; c++) q0[c] = src0[7 + c];
          }
          ref0_done = true;
        }
        __syncthreads();                   // the hand-over area overlaps the state-row image: everyone is done reading it
        lds_double* wg_lds = regions + kHO;
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
          const int c = 16 * (ct0 + ct) + c16;
          const int vb = h7 ? c / 7 : c / 11, col = c - ncv * vb;
          if (c < ncols && (ct < 2 || !h7)) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
              if (PACK)   // node 16 (w & 1) + .. of vector vb -> lane 32 (vb & 1) + node of wavefront vb >> 1
                wg_lds[(vb >> 1) * kWL + (32 * (vb & 1) + 16 * (wv & 1) + kq + 4 * i) * 11 + col] = acc[ct][i];
              else
                wg_lds[vb * kWL + (16 * wv + kq + 4 * i) * 11 + col] = acc[ct][i];
            }
          }
        }
        __syncthreads();
        if (ghost) return;
        lds_double* row = wave_lds + kHO + lane * 11;
        lm = row[0];
#pragma unroll
        for (int c = 0; c < 3; c++) { lr[c] = row[1 + c]; lv[c] = row[4 + c]; }
#pragma unroll
        for (int c = 0; c < 4; c++) lq[c] = row[7 + c];
      } else if (XLDS) {
        // Longer phases (kXldsPipeFrom nodes and more): the product in slabs of 32 state rows (8 k-steps) with BOTH operands
        // brought into LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) --
        //   * the state rows of slab s + 1, per vector as they lie in x, [mass R | position 3R | velocity 3R | quaternion 4R]
        //     (R = 32 rows: 176 contiguous 16-byte pieces, three instructions per wavefront), into the other of two images
        //     while slab s is multiplied;
        //   * the A operands in a ring of one slab's k-steps (this wavefront's row tile, 512 B per k-step, two k-steps per
        //     instruction): a pair of slots is refilled with the next slab's values as soon as it has been multiplied, a whole
        //     slab before it is needed.
        // Vector-memory results return in order; the register-staged form before this one requested the A operand one k-step
        // ahead, behind the next slab's state rows (HBM), and the compiler -- unable to count across the lane-conditional
        // staging loads -- waited for every outstanding load before every k-step (upper bound of those stalls at 12 x 128:
        // 21 % of the launch).  Here nothing the compiler tracks is in flight: the waits are counted by hand (queue order:
        // refills of the previous slab, three row pieces, refills of this slab), the barrier is the bare instruction.
        const int c16 = lane & 15, kq = lane >> 4;
        constexpr int kPipeK = kPipeSlabK, kPipeRows = 4 * kPipeK, kPipeBuf = kPipeRows * 11;
        constexpr int kRingOff = 2 * kPipeBuf, kPairs = kPipeK / 2;
        static_assert(kPipeK == 8, "the piece tables below are written for 32-row slabs");
        static_assert(!COOP || PACK || !LONGP || kRingOff + 64 * kPipeK <= kWL, "two images and the A ring must fit the wave's region");
        constexpr int kBP = kPipeRows, kBV = 4 * kPipeRows, kBQ = 7 * kPipeRows;   // blocks of an image
        lds_double* regions = (lds_double*)lds + park_off;
        const bool h7 = GEL_DX_HOLD7 && ph.hold;   // hold-type phase: seven columns per vector in two column tiles (see the one-slab form)
        const int ncv = h7 ? 7 : 11, ncols = 4 * ncv;
        int xoff[3], xstr[3];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
          const int c = 16 * ct + c16;
          const int vb = min(h7 ? c / 7 : c / 11, 3), col = (c < ncols) ? c - ncv * vb : 0;
          const int base = (col == 0) ? 0 : ((col < 4) ? kBP + col - 1 : ((col < 7) ? kBV + col - 4 : kBQ + col - 7));
          const int stride = (col == 0) ? 1 : ((col < 7) ? 3 : 4);
          xoff[ct] = vb * kWL + base + stride * kq;      // row kq of the lane's column
          xstr[ct] = stride;                             // doubles from a row of the column to the next
        }
        gel_double4 acc[3];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) acc[ct] = gel_double4{0.0, 0.0, 0.0, 0.0};
        typedef const __attribute__((address_space(1))) void* gel_gptr;
        typedef __attribute__((address_space(3))) void* gel_lptr;
        // piece tables: 16-byte piece 64 i + lane of an image <- doubles s_i + slab * a_i of this vector
        const int xa0 = ph.xa;
        const int s0 = (lane < 16) ? xa0 + 2 * lane : M + 3 * xa0 + 2 * (lane - 16);
        const int s1 = (lane < 48) ? 4 * M + 3 * xa0 + 2 * lane : 7 * M + 4 * xa0 + 2 * (lane - 48);
        const int s2 = 7 * M + 4 * xa0 + 2 * (16 + lane);
        const int a0 = (lane < 16) ? kPipeRows : 3 * kPipeRows, a1 = (lane < 48) ? 3 * kPipeRows : 4 * kPipeRows, a2 = 4 * kPipeRows;
#define GEL_PIPE_ROWS(slab, buf)                                                                                   \
  do {                                                                                                             \
    lds_double* _img = wave_lds + (buf) * kPipeBuf;                                                                \
    __builtin_amdgcn_global_load_lds((gel_gptr)(xb + s0 + (slab) * a0), (gel_lptr)_img, 16, 0, 0);                 \
    __builtin_amdgcn_global_load_lds((gel_gptr)(xb + s1 + (slab) * a1), (gel_lptr)(_img + 128), 16, 0, 0);         \
    if (lane < 48) __builtin_amdgcn_global_load_lds((gel_gptr)(xb + s2 + (slab) * a2), (gel_lptr)(_img + 256), 16, 0, 0); \
  } while (0)
        // A operands: Dst[((item + k) * 4 + row tile) * 64 + lane]; lanes 0..31 fetch k-step k, lanes 32..63 k-step k + 1
        const double* ap2 = P.Dst + (size_t)dsw * 4 + wv * 64 + (lane & 31) * 2;
        const int khalf = lane >> 5, kmax = ((n + 4) >> 2) - 1;          // the host lays down (n + 4) / 4 k-steps per work item
#define GEL_PIPE_A(pair, k)                                                                                        \
  __builtin_amdgcn_global_load_lds((gel_gptr)(ap2 + (size_t)min((k) + khalf, kmax) * 256), (gel_lptr)(wave_lds + kRingOff + 128 * (pair)), 16, 0, 0)
#define GEL_PIPE_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
        GEL_PIPE_ROWS(0, 0);
#pragma unroll
        for (int pr = 0; pr < kPairs; pr++) GEL_PIPE_A(pr, 2 * pr);
        stage_tables_commit(P, lds, tab_mine);
        GEL_PIPE_WAIT(0);
        __syncthreads();
        const int nslab = (((n + 4) >> 2) + kPipeK - 1) / kPipeK;   // slabs that hold rows 0 .. n (the tail row and the last node's own row included)
        const int own = jc + 1;
        const lds_double* ring = wave_lds + kRingOff + lane;
        for (int sl = 0; sl < nslab; sl++) {
          const int bo = (sl & 1) * kPipeBuf;
          const bool more = sl + 1 < nslab;  /* wave-uniform */
          const int k0 = sl * kPipeK;
          if (more) GEL_PIPE_ROWS(sl + 1, (sl + 1) & 1);  /* in flight while this slab multiplies */
          int p0 = xoff[0] + bo, p1 = xoff[1] + bo, p2 = xoff[2] + bo;  /* running operand addresses (not 24 hoisted ones) */
#pragma unroll
          for (int pr = 0; pr < kPairs; pr++) {
            /* this pair's refill was the pr-th of the previous slab: behind it in the queue are that slab's later refills */
            /* (kPairs - 1 - pr), this slab's three row pieces and its earlier refills (pr) -- or, in the last slab, only the former */
            if (more) GEL_PIPE_WAIT(kPairs + 2); else GEL_PIPE_WAIT(kPairs - 1 - pr);
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int ks = 2 * pr + h;
              if (k0 + ks < ksteps) {  /* wave-uniform */
                const double a = ring[64 * ks];
                const double bl0 = regions[p0], bl1 = regions[p1];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl1, acc[1], 0, 0, 0);
                if (!h7) {
                  const double bl2 = regions[p2];
                  acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl2, acc[2], 0, 0, 0);
                }
              }
              p0 += 4 * xstr[0]; p1 += 4 * xstr[1]; p2 += 4 * xstr[2];
              asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));  /* keep them running */
            }
            if (more) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* the pair's slots have been read */
              GEL_PIPE_A(pr, k0 + kPipeK + 2 * pr);
            }
          }
          {
            const int ol = own - sl * kPipeRows;  /* the node's own state row, if it lies in this slab */
            if (ol >= 0 && ol < kPipeRows) {
              const lds_double* img = wave_lds + bo;
              me = img[ol];
#pragma unroll
              for (int c = 0; c < 3; c++) { re[c] = img[kBP + 3 * ol + c]; ve[c] = img[kBV + 3 * ol + c]; }
#pragma unroll
              for (int c = 0; c < 4; c++) q[c] = img[kBQ + 4 * ol + c];
            }
          }
          if (n >= sl * kPipeRows && n < (sl + 1) * kPipeRows)  /* wave-uniform: the tail row lies in this slab's image */
            GEL_DX_TAIL_ACC(h7 ? 2 : 3, regions[xoff[ct] + bo + (n - sl * kPipeRows - kq) * xstr[ct]]);
          if (more) {
            GEL_PIPE_WAIT(kPairs);  /* the next slab's rows have landed (behind them: this slab's refills) */
            if (!tail1 && sl + 2 == nslab && lane < 3) {
              /* rows past n of the last k-step multiply columns of D that hold zeros: they must not hold another phase's numbers */
              /* (a NaN there would reach this phase's rows) */
              const int rz = n + 1 + lane - (sl + 1) * kPipeRows;
              if (rz < kPipeRows && n + 1 + lane < 4 * ksteps) {
                lds_double* img = wave_lds + ((sl + 1) & 1) * kPipeBuf;
                img[rz] = 0.0;
#pragma unroll
                for (int c = 0; c < 3; c++) { img[kBP + 3 * rz + c] = 0.0; img[kBV + 3 * rz + c] = 0.0; }
#pragma unroll
                for (int c = 0; c < 4; c++) img[kBQ + 4 * rz + c] = 0.0;
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();  /* slab sl is consumed everywhere, slab sl + 1 is in place */
        }
#undef GEL_PIPE_ROWS
#undef GEL_PIPE_A
#undef GEL_PIPE_WAIT
        lds_double* wg_lds = regions + kHO;
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
          const int c = 16 * ct + c16;
          const int vb = h7 ? c / 7 : c / 11, col = c - ncv * vb;
          if (c < ncols && (ct < 2 || !h7)) {
#pragma unroll
            for (int i = 0; i < 4; i++) wg_lds[vb * kWL + (16 * wv + kq + 4 * i) * 11 + col] = acc[ct][i];
          }
        }
        __syncthreads();
        if (ghost) return;
        lds_double* row = wave_lds + kHO + lane * 11;
        lm = row[0];
#pragma unroll
        for (int c = 0; c < 3; c++) { lr[c] = row[1 + c]; lv[c] = row[4 + c]; }
#pragma unroll
        for (int c = 0; c < 4; c++) lq[c] = row[7 + c];
      } else if (MFMA && csplit) {
        // SPLIT, whole one-vector evaluation: this wavefront's row tile (rows 16 part .. + 15) of [64 x (n+1)] . [(n+1) x 11]; A operands
        // row-tile major (ProblemDev::Dst, 512 B per k-step), B as in the branch below.
        const int c16 = lane & 15, kq = lane >> 4;
        const double* bp = xm + ph.xa;
        int bs = 1;
        if (c16 >= 1 && c16 < 4) { bp = xr + 3 * ph.xa + (c16 - 1); bs = 3; }
        if (c16 >= 4 && c16 < 7) { bp = xv + 3 * ph.xa + (c16 - 4); bs = 3; }
        if (c16 >= 7) { bp = xq + 4 * ph.xa + ((c16 < 11) ? (c16 - 7) : 0); bs = 4; }
        gel_double4 acc1 = gel_double4{0.0, 0.0, 0.0, 0.0};
        const double* ap1 = P.Dst + (size_t)dsw * 4 + part * 64 + lane;
        const unsigned un = (unsigned)n, ubs = (unsigned)bs;
        if (tail1 && rb) {   // wave-uniform addresses (the lead adds the last state row after the hand-over)
          xl[0] = xm[ph.xa + n];
#pragma unroll
          for (int c = 0; c < 3; c++) { xl[1 + c] = xr[3 * (ph.xa + n) + c]; xl[4 + c] = xv[3 * (ph.xa + n) + c]; }
#pragma unroll
          for (int c = 0; c < 4; c++) xl[7 + c] = xq[4 * (ph.xa + n) + c];
        }
        constexpr int kSplitB = SPLITB, kSplitA = 4;
        if (n < kSlabRowsMax) {
          // phases of at most 67 nodes: the state rows cross the bus ONCE -- every wavefront fetches a quarter of the [68][11] image
          // (x of a one-vector call sits in pinned HOST memory, which no cache holds: four wavefronts each reading every row was
          // four times the PCIe traffic), into the region of part 1; one more barrier, then the B operands come from LDS
          static_assert(!(SPLIT && MFMA && !JAC) || kSlabRowsMax * 11 <= kWL, "the state-row image must fit a wavefront's region");
          lds_double* img = wave_lds + (1 - part) * kWL;
          constexpr int kQRows = kSlabRowsMax / 4, kQ = kQRows * 11;   // 17 rows = 187 values per wavefront
#pragma unroll
          for (int i = 0; i < (kQ + 63) / 64; i++) {
            const int e = lane + 64 * i;
            const int rl = (e * 373) >> 12, c = e - 11 * rl;           // e / 11, e % 11 for e < 192
            const int r = kQRows * part + rl;
            if (e < kQ) {
              const int xr_ = ph.xa + min(r, n);
              const int off = (c == 0) ? xr_ : ((c < 4) ? M + 3 * xr_ + (c - 1) : ((c < 7) ? 4 * M + 3 * xr_ + (c - 4) : 7 * M + 4 * xr_ + (c - 7)));
              const double v = xb[off];
              img[r * 11 + c] = (r <= n) ? v : 0.0;                   // rows past the phase meet zero columns of D: they hold zeros
            }
          }
          double a1[kSplitA];
#pragma unroll
          for (int i = 0; i < kSplitA; i++) a1[i] = ap1[min(i, ksteps - 1) * 256];
          __syncthreads();
          const int bo = kq * 11 + ((c16 < 11) ? c16 : 0);
#pragma unroll
          for (int i = 0; i < kSlabRowsMax / 4; i++) {
            if (i < ksteps) {   // wave-uniform
              acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i % kSplitA], img[bo + 44 * i], acc1, 0, 0, 0);
              a1[i % kSplitA] = ap1[min(i + kSplitA, ksteps - 1) * 256];
            }
          }
        } else {
          for (int k0 = 0; k0 < ksteps; k0 += kSplitB) {
            double bl[kSplitB], a1[kSplitA];
#pragma unroll
            for (int i = 0; i < kSplitB; i++) bl[i] = bp[min(4u * (k0 + i) + kq, un) * ubs];
#pragma unroll
            for (int i = 0; i < kSplitA; i++) a1[i] = ap1[min(k0 + i, ksteps - 1) * 256];
#pragma unroll
            for (int i = 0; i < kSplitB; i++) {
              if (k0 + i < ksteps) {   // wave-uniform
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i % kSplitA], bl[i], acc1, 0, 0, 0);
                a1[i % kSplitA] = ap1[min(k0 + i + kSplitA, ksteps - 1) * 256];
              }
            }
          }
        }
        // the tile into the LEAD's staging tile (the lead is wavefront `part` places before this one), one barrier, the lead takes its rows
        lds_double* lead_lds = wave_lds - part * kWL;
#pragma unroll
        for (int i = 0; i < 4; i++) lead_lds[(16 * part + kq + 4 * i) * kStageLd + c16] = acc1[i];
        __syncthreads();
        if (only_dx) return;
        if (rb) {
          lds_double* row = wave_lds + lane * kStageLd;
          lm = row[0];
#pragma unroll
          for (int c = 0; c < 3; c++) { lr[c] = row[1 + c]; lv[c] = row[4 + c]; }
#pragma unroll
          for (int c = 0; c < 4; c++) lq[c] = row[7 + c];
        }
      } else if (MFMA) {
        // SPLIT (latency form): one wavefront multiplies alone.
        // [64 x (n+1)] . [(n+1) x 11] per wavefront as 4 row tiles of v_mfma_f64_16x16x4_f64.
        // Operand layout (cdna_hip_programming.md section 3): A lane l = A[row l&15][k l>>4],
        // B lane l = B[k l>>4][col l&15], C/D reg i of lane l = C[row (l>>4)+4i][col l&15].
        const int c16 = lane & 15, kq = lane >> 4;
        // column c16 of X = one of the 11 interleaved state columns (mass | pos xyz | vel xyz | quat wxyz)
        const double* bp = xm + ph.xa;
        int bs = 1;
        if (c16 >= 1 && c16 < 4) { bp = xr + 3 * ph.xa + (c16 - 1); bs = 3; }
        if (c16 >= 4 && c16 < 7) { bp = xv + 3 * ph.xa + (c16 - 4); bs = 3; }
        if (c16 >= 7) { bp = xq + 4 * ph.xa + ((c16 < 11) ? (c16 - 7) : 0); bs = 4; }
        gel_double4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = gel_double4{0.0, 0.0, 0.0, 0.0};
        // A operands come pre-arranged (ProblemDev::Dsw): one 32-byte access per lane and k-step fetches the
        // values of all four row tiles; columns past n hold zeros there, so B is only clamped, never masked.
        const gel_double4* ap = reinterpret_cast<const gel_double4*>(P.Dsw) + (size_t)dsw * 1 + lane;
        const unsigned un = (unsigned)n, ubs = (unsigned)bs;
        if (tail1) {   // wave-uniform addresses
          xl[0] = xm[ph.xa + n];
#pragma unroll
          for (int c = 0; c < 3; c++) { xl[1 + c] = xr[3 * (ph.xa + n) + c]; xl[4 + c] = xv[3 * (ph.xa + n) + c]; }
#pragma unroll
          for (int c = 0; c < 4; c++) xl[7 + c] = xq[4 * (ph.xa + n) + c];
        }
        // Latency form: the round trips are what counts (B = 1: x sits in pinned HOST memory, a load is a PCIe read).  The state
        // column of the first kSplitB k-steps (a 64-node phase: all 17) is requested at once, the A operands run kSplitA
        // k-steps ahead (L2), instead of one dependent round trip per k-step; longer phases go on in blocks of kSplitB k-steps.
        constexpr int kSplitB = SPLITB, kSplitA = 2;
        for (int k0 = 0; k0 < ksteps; k0 += kSplitB) {
          double bl[kSplitB];
          gel_double4 a4[kSplitA];
#pragma unroll
          for (int i = 0; i < kSplitB; i++) bl[i] = bp[min(4u * (k0 + i) + kq, un) * ubs];
#pragma unroll
          for (int i = 0; i < kSplitA; i++) a4[i] = ap[min(k0 + i, ksteps - 1) * 64];
#pragma unroll
          for (int i = 0; i < kSplitB; i++) {
            if (k0 + i < ksteps) {   // wave-uniform
#pragma unroll
              for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[i % kSplitA][t], bl[i], acc[t], 0, 0, 0);
              a4[i % kSplitA] = ap[min(k0 + i + kSplitA, ksteps - 1) * 64];
            }
          }
        }
        // transpose through the wave's LDS region: tile -> one row (node) per lane
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
          for (int i = 0; i < 4; i++) wave_lds[(16 * t + kq + 4 * i) * kStageLd + c16] = acc[t][i];
        lds_double* row = wave_lds + lane * kStageLd;
        lm = row[0];
#pragma unroll
        for (int c = 0; c < 3; c++) { lr[c] = row[1 + c]; lv[c] = row[4 + c]; }
#pragma unroll
        for (int c = 0; c < 4; c++) lq[c] = row[7 + c];
      } else {
        // lane j reads consecutive Dt addresses; the X rows are wave-uniform -> scalar loads feeding
        // v_fma_f64 as SGPR operands
        const double* Dt = P.Dt + ph.doff + j;
        const double* pm = xm + ph.xa;
        const double* pr = xr + 3 * ph.xa;
        const double* pv = xv + 3 * ph.xa;
        const double* pq = xq + 4 * ph.xa;
        for (int i = 0; i <= n; i++) {
          const double d = Dt[(size_t)i * n];
          lm += d * pm[i];
#pragma unroll
          for (int c = 0; c < 3; c++) lr[c] += d * pr[3 * i + c];
#pragma unroll
          for (int c = 0; c < 3; c++) lv[c] += d * pv[3 * i + c];
#pragma unroll
          for (int c = 0; c < 4; c++) lq[c] += d * pq[4 * i + c];
        }
      }
    }
    GEL_TAKE_KNOT_TIMES();
#undef GEL_DX_TAIL_ACC
    if (SPLIT && tail1 && rb) {   // latency form, D.X: row n of the phase (see tail1 above)
      lm = __builtin_fma(dlast, xl[0], lm);
#pragma unroll
      for (int c = 0; c < 3; c++) { lr[c] = __builtin_fma(dlast, xl[1 + c], lr[c]); lv[c] = __builtin_fma(dlast, xl[4 + c], lv[c]); }
#pragma unroll
      for (int c = 0; c < 4; c++) lq[c] = __builtin_fma(dlast, xl[7 + c], lq[c]);
    }
    if (MFMA && !active) return;  // ragged tail: nothing to write
    // the staging tile has been consumed: the region now becomes the park of what the velocity group
    // needs late (its sweeps re-read quaternion, velocity and D[j][j+1]; its defect needs the D.X row)
    PARK_SET(PK_Q0, q[0]); PARK_SET(PK_Q1, q[1]); PARK_SET(PK_Q2, q[2]); PARK_SET(PK_Q3, q[3]);
    PARK_SET(PK_V0, ve[0]); PARK_SET(PK_V1, ve[1]); PARK_SET(PK_V2, ve[2]);
    if (JAC) PARK_SET(PK_DJJ, djj);
    if (rb) { PARK_SET(PK_LV0, lv[0]); PARK_SET(PK_LV1, lv[1]); PARK_SET(PK_LV2, lv[2]); }
    // last global loads: the reference rows of engine-off / hold phases, in the forms without an image of the state rows
    if (!ref0_done) {
      if (rb && !ph.engine_on) m0 = xm[ph.xa];
      if (rb && ph.hold) {
#pragma unroll
        for (int c = 0; c < 4; c++) q0[c] = xq[4 * ph.xa + c];
      }
    }

    // ---- everything that needs no velocity RHS is finished here, while its inputs are in registers:
    //      position Jacobian entries, the whole quaternion group (:155-213, :499-632) ----
    if (JAC && lead) {
      // pos/velocity diagonal (:190-196): the same value for every node and component -> one scalar per phase
      if (coo) {   // the 3 n copies of the scalar in pos/velocity (:180-183), this chunk's share
        const double sc = -(P.kpt * (tf - to));
        double* d_ = P.coo_full + GEL_COO_BASE(CG_PV) + (size_t)3 * j0;
#pragma unroll
        for (int i = 0; i < 3; i++) d_[(cw8 == 512) ? 64 * i + lane : 3 * lane + i] = sc;   // ragged chunk: only the lanes with a node are alive
      } else if (j == 0) EMIT_AT(packed ? (ph.K - sub_hi) * cw8 : ph.K * n * 8, -(P.kpt * (tf - to)));   // behind all chunks' blocks of the phase (packed: behind the unit's slots)
      {
        const double pt[3] = {ve[0] * P.kpt, ve[1] * P.kpt, ve[2] * P.kpt};   // t0 column; tf = its negative
        EMIT_GROUP(3, kSlotPT, pt, CG_PT, 0, true, true);
      }
    }
    if (!ph.hold && lead) {
      double fq[4];
      quat_rate(q, u0, u1, P.uu, fq);
      if (rb) {
        double cq[4];
#pragma unroll
        for (int c = 0; c < 4; c++) cq[c] = lq[c] - fq[c] * (tf - to) * hT;
        RSTORE_ROWS(wave_lds + kTileOff, 4, rs_q, cq);   // ordinary (not non-temporal) stores: non-temporal ones cost 4 % more HBM writes (PMC, round 2)
      }
      if (JAC && ph.q_fd) {
        double f[4];
        // GEL_FLAG_FD_RECOMPUTE: the reference's sweeps.
        // submat_quat[4j+c, 4(j+1)+k] = D[j][j+1]*(c==k) + rh_quat   (con_dynamics.py:575-589).  dq_c contains q_k only
        // for (c, k) in {0,1} x {2,3} and {2,3} x {0,1}: the other eight differences are exactly zero in the reference
        // too (the perturbed component never enters), so those entries are constants of the pattern.
#pragma unroll
        for (int k = 0; k < 4; k++) {
          double qp[4];
#pragma unroll
          for (int c = 0; c < 4; c++) qp[c] = (k == c) ? (q[c] + dx) : q[c];
          quat_rate(qp, u0, u1, P.uu, f);
          const int c0 = (k < 2) ? 2 : 0;
          EMIT(ph.s_qq + 2 * k, FDQ(f[c0], fq[c0]));
          EMIT(ph.s_qq + 2 * k + 1, FDQ(f[c0 + 1], fq[c0 + 1]));
        }
#pragma unroll 1
        for (int k = 0; k < 2; k++) {
          quat_rate(q, (k == 0) ? u0 + dx : u0, (k == 1) ? u1 + dx : u1, P.uu, f);
#pragma unroll
          for (int c = 0; c < 4; c++) EMIT(ph.s_qq + 8 + 4 * k + c, FDQ(f[c], fq[c]));
        }
#pragma unroll
        for (int c = 0; c < 4; c++) EMIT(ph.s_qq + 16 + c, fq[c] * hT);  // t0 column; tf = its negative
      } else if (JAC) {
        // dq = q (x) (0, 0, omega_y, omega_z) / 2 is linear in q and in u (src/pybind_dynamics.cpp:94-106), so the reference's
        // difference quotients ARE its partial derivatives up to the rounding of two evaluations (1e-8 of |dq|): the sixteen
        // entries of quat / quaternion and quat / u are +- six numbers -- omega_y S / 2, omega_z S / 2 and
        // unit_u (pi / 180) q_i S / 2 with S = (tf - to) unit_t / 2 -- which the gather map signs and places (gel_host.hip).
        const double hS = (tf - to) * ut / 2.0;
        const double d2r = 0.017453292519943295769;
        EMIT(ph.s_qq + 0, 0.5 * ((u0 * P.uu) * d2r) * hS);
        EMIT(ph.s_qq + 1, 0.5 * ((u1 * P.uu) * d2r) * hS);
        const double kq = 0.5 * (P.uu * d2r) * hS;
        if (coo) {
          // quat/u (:600-613): entry (c, k) = -d(dq_c)/d(u_k) S = sign[k][c] C_which[k][c], C_i = kq q_i -- what the gather map makes of the
          // four slots below (gel_host.hip walk_pattern), spelled out: the same products, the same bits
          const double C0 = kq * q[0], C1 = kq * q[1], C2 = kq * q[2], C3 = kq * q[3];
          const double u0v[4] = {C2, C3, -C0, -C1}, u1v[4] = {C3, -C2, C1, -C0};
          EMIT_GROUP(4, 0, u0v, CG_QU, 0, false, true);
          EMIT_GROUP(4, 0, u1v, CG_QU, 1, false, true);
        } else {
          EMIT(ph.s_qq + 2, -(kq * q[0]));
          EMIT(ph.s_qq + 3, kq * q[1]);
          EMIT(ph.s_qq + 4, kq * q[2]);
          EMIT(ph.s_qq + 5, kq * q[3]);
        }
        {
          const double qt[4] = {fq[0] * hT, fq[1] * hT, fq[2] * hT, fq[3] * hT};   // t0 column; tf = its negative
          EMIT_GROUP(4, ph.s_qq + 6, qt, CG_QT, 0, true, true);
        }
      }
    }
    if (rb) {
      // ---- mass, position (and hold-type quaternion) defects (:34-63,116-152,521-522) ----
      double cm;
      if (ph.engine_on) {
        const double rh = ph.mf_um * (tf - to) * ut / 2.0;
        cm = lm - rh;
      } else {
        cm = me - m0;
      }
      RSTORE(rs_m + gn, cm);
      GEL_CHK(cm);
      {
        double cp[3];
        const double kpos = P.kpt * (tf - to);   // unit_v (tf - to) unit_t / 2 / unit_p, wave-uniform
#pragma unroll
        for (int c = 0; c < 3; c++) cp[c] = lr[c] - ve[c] * kpos;
        RSTORE_ROWS(wave_lds + kTileOff, 3, rs_p, cp);
      }
      if (ph.hold) {
        double cq[4];
#pragma unroll
        for (int c = 0; c < 4; c++) cq[c] = q[c] - q0[c];
        RSTORE_ROWS(wave_lds + kTileOff, 4, rs_q, cq);
      }
    }
  }
  // ======================= from here on: no global loads =======================
  // compiler barrier: parked values are re-read from LDS below, not forwarded through VGPRs
  asm volatile("" ::: "memory");

  // ---------------- velocity RHS + FD Jacobian (lib/con_dynamics.py:216-496) ----------------
  double fc[3];
  {
    const double tn = tau * (tf - to) / 2 + (tf + to) / 2;  // PSparams.time_nodes, SectionParameters.py:77-81
    const double inv_m = frcp(me * P.um);

    // Exact-difference forms of the two sweeps of the velocity group whose perturbed variable enters the RHS algebraically
    // (the reference differences two runs, lib/con_dynamics.py:402-450; GEL_FLAG_FD_RECOMPUTE keeps that):
    //  * quaternion component k: the thrust direction is a quadratic form of q (thrust_dir()), so with the step the
    //    reference's `+= dx` really takes, dl = (q_k + dx) - q_k (exact), its change is dl * (d dir / d q_k) plus dl^2 in the first
    //    component -- f_p - f_c = T / m / unit_v * that, exactly; 11 operations per sweep instead of a thrust direction, an
    //    acceleration and three differences;
    //  * mass: f_c - f_p = (T d + F) (1/m - 1/m') / unit_v = tm * e / (1 + e) / unit_v with e = (m' - m) / m from the two
    //    rounded products the reference forms (their difference is exact); e <= 1e-6, so e (1 - e (1 - e)) is exact to 1e-18.
#define GEL_QUAT_CLOSED(Tval)                                                                                          \
  do {                                                                                                                 \
    const double kq_ = ((Tval) * inv_m) * (inv_uv * fds);                                                              \
    const double q_[4] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2), PARK_GET(PK_Q3)};                         \
    _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                                    \
      const double dl_ = (q_[k] + dx) - q_[k];                                                                         \
      const double a_ = (k < 2) ? -(dl_ * kq_) : dl_ * kq_;   /* entry = -(f_p - f_c) fds; d dir_x / d q_k = +-2 q_k */ \
      const double b_ = -2.0 * (dl_ * kq_);                                                                            \
      const double vq_[3] = {(2.0 * q_[k] + dl_) * a_,                                                                 \
                             q_[3 - k] * b_,                  /* d dir_y / d q = 2 (q3, q2, q1, q0) */                  \
                             ((k & 1) ? q_[k ^ 2] : -q_[k ^ 2]) * b_};   /* d dir_z / d q = 2 (-q2, q3, -q0, q1) */     \
      EMIT_GROUP(3, ph.s_vq + 3 * k, vq_, CG_VQ, k, false, true);                                                      \
    }                                                                                                                  \
  } while (0)
#define GEL_MASS_CLOSED(tm_)                                                                                           \
  do {                                                                                                                 \
    /* two ROUNDED products, as the reference forms them (a contracted fma would difference an unrounded one) */      \
    const double e_ = (fresh_product(me + dx, P.um) - fresh_product(me, P.um)) * inv_m;                                \
    const double k_ = (e_ * (1.0 - e_ * (1.0 - e_))) * (inv_uv * fds);                                                 \
    const double vm_[3] = {(tm_)[0] * k_, (tm_)[1] * k_, (tm_)[2] * k_};                                               \
    EMIT_GROUP(3, kSlotVM, vm_, CG_VM, 0, false, true);                                                                \
  } while (0)
    if (ph.air) {
      // The Earth angle omega t enters the RHS only through the rotation of the wind into ECI (the air-relative velocity's two
      // rotations cancel, aero_force()): a wavefront in calm air -- both wind components exactly zero in every lane, e.g. above
      // and below the measured part of the wind table -- never needs it.  It is formed on first need (wave-uniform), its
      // half-angle pair kept (full_angle()); position sweeps do not change it.
      EarthHalf eh{1.0, 0.0};
      bool have_eh = false;
      // the node's time once more, for the rare late first need (a load behind the stores: it waits for them)
#define GEL_NEED_EARTH_ANGLE(wn_, we_)                                                                                 \
  do {                                                                                                                 \
    if (!have_eh && __builtin_amdgcn_ballot_w64(!((wn_) == 0.0 && (we_) == 0.0)) != 0) {                                \
      int jl_ = jc;                                                                                                    \
      if (AERO) asm volatile("" : "+v"(jl_));   /* AERO: the address is formed here, not carried from the top of the kernel */ \
      const double tau_ = P.tau[ph.toff + jl_];                                                                        \
      const EarthAngle e_ = earth_angle(tau_ * (tf - to) / 2 + (tf + to) / 2);                                         \
      eh.ch = e_.ch; eh.sh = e_.sh; have_eh = true;                                                                    \
    }                                                                                                                  \
  } while (0)
      // Order of this branch (register and park discipline; the kernel runs 4 waves/SIMD on 128 VGPRs and a spilled value
      // would be reloaded through vmcnt, i.e. behind every Jacobian store in flight):
      // (1) the centre evaluation; the intermediates of its position part that the position sweeps will need (PosCentre) go to
      //     the park as soon as they exist -- FP0-7 at once, LV0-2 once the velocity defect has been stored;
      // (2) the light sweeps (velocity, quaternion, mass) and the t columns, while the centre's wind, force, gravity and
      //     thrust are in registers, where they then die; the centre value f_c and the thrust direction go to Q0-3 / DJJ;
      // (3) the three position sweeps in exact-difference form (pos_delta(): the change of altitude, atmosphere and wind from
      //     algebraic identities, ~130 operations instead of the ~450 of a second run of the chain), reading PosCentre back
      //     from the park, each sweep's three entries written at once;
      // (4) a wavefront with a lane the difference form does not cover (perturbed point in another atmosphere layer or table
      //     piece, node next to the polar axis), and every wavefront of a problem created with GEL_FLAG_FD_RECOMPUTE, re-runs
      //     the chain on the perturbed position like the reference does;
      // (5) GEL_FLAG_FD_RECOMPUTE only: the t0 / tf sweeps.
      // The scaled position r = re * unit is formed where it is used (fresh_product) instead of living next to re.
      PosCentreTail pt;
      double cen_rho = 0.0, cen_P = 0.0, cen_inv_a = 0.0;   // the centre's density, pressure, 1 / speed of sound
      double dir2 = 0.0;   // z component of the thrust direction (x, y are parked)
      // AERO: what the aero rows of the position sweeps need of the centre -- cos(alpha), alpha, q (parked; 1 / sin(alpha) is formed
      // again; whether alpha's difference form applies at the centre rides in the SIGN of the parked alpha, which is never negative
      // itself: acos or the clamp's +0); the Earth angle of con_aero.py, which takes the node's
      // time in SECONDS (lib/con_aero.py:45; the defect's right-hand side takes the normalised time, lib/con_dynamics.py:246):
      // half-angle pair, formed on first need like the defect's
      EarthHalf aeh{1.0, 0.0};
      bool have_aeh = false;
#ifndef GEL_AEH
#define GEL_AEH aeh
#endif
#ifndef GEL_X_NOPOS
#define GEL_X_NOPOS 0
#endif
#ifndef GEL_X_NOQUAT
#define GEL_X_NOQUAT 0
#endif
#ifndef GEL_X_NOVEL
#define GEL_X_NOVEL 0
#endif
      const bool aero_on = AERO && akinds != 0;                 // wave-uniform
      const bool a_need_alpha = AERO && (akinds & 5) != 0;      // an alpha or q-alpha row
#define GEL_AERO_NEED_EA(wn_, we_)                                                                                     \
  do {                                                                                                                 \
    if (!have_aeh && __builtin_amdgcn_ballot_w64(!((wn_) == 0.0 && (we_) == 0.0)) != 0) {                               \
      int jl_ = jc;                                                                                                    \
      asm volatile("" : "+v"(jl_));   /* the address is formed here, not carried (and spilled) from the top of the kernel */ \
      const double tau_ = P.tau[ph.toff + jl_];                                                                        \
      const EarthAngle e_ = earth_angle((tau_ * (tf - to) / 2 + (tf + to) / 2) * P.ut);                                \
      aeh.ch = e_.ch; aeh.sh = e_.sh; have_aeh = true;                                                                 \
    }                                                                                                                  \
  } while (0)
      // One gradient entry of every kind of this phase, -(f_p - f_c) / dx / limit (lib/con_aero.py:437-463), from the perturbed
      // point's air-relative velocity a_ (squared norm nv2_), body axis d_ (1 / |d| = ind_) and density: aero_body's GEL_AERO_EMIT
      // (gel_kernels.hip), expression for expression.  blk: 0 position, 1 velocity, 2 quaternion; okl_: this lane writes.
      // A kind this phase does not have (akinds) is not branched around: its block base points at the record's dump area and its
      // scale factors are zero (gel_host.hip) -- three scalar tests and branches less per entry, nothing written where it counts.
      // the phase's aero record, fetched ONCE per group of entries (scalar loads, one round trip -- fetched where each store needs
      // them they were 35 round trips per wavefront, each waited for on the spot): block bases, scale factors; n8_ = bytes of a
      // column's row, j08_ = this chunk's first node in it
#define GEL_AERO_FETCH()                                                                                               \
  const int ab_[3] = {load_const(&aph->base[0]), load_const(&aph->base[1]), load_const(&aph->base[2])};                \
  const double ax_[3] = {load_const(&aph->ilx[0]), load_const(&aph->ilx[1]), load_const(&aph->ilx[2])};                \
  const int n8_ = 8 * n, j08_ = 8 * j0
#define GEL_AERO_ENTRY(blk, col, a_, nv2_, d_, ind_, rho_, okl_, a_cc, a_ac, a_qc, a_isc, a_ok)                         \
  do {                                                                                                                 \
    double t_ = 0.0, dq_ = 0.0, qp_ = 0.0;                                                                             \
    const double ac_ = a_ac;                                                                                           \
    {                                                                                                                  \
      const double qc_ = a_qc;                                                                                         \
      qp_ = 0.5 * (rho_) * (nv2_);                                                                                     \
      dq_ = qp_ - qc_;                                                                                                 \
      if (a_need_alpha) {                                                                                              \
        const double cp_ = aero_cos(a_, nv2_, d_, ind_);                                                               \
        const bool okd_ = aero_dalpha(cp_, nv2_, a_cc, a_isc, a_ok, t_);                                               \
        if (__builtin_amdgcn_ballot_w64(!okd_) != 0) {                                                                 \
          const double t2_ = aero_acos(cp_, nv2_) - ac_;                                                               \
          t_ = okd_ ? t_ : t2_;                                                                                        \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    _Pragma("unroll") for (int kind = 0; kind < 3; kind++) {                                                           \
      if ((blk) == 2 && kind == 1) continue;               /* dynamic pressure has no quaternion block */              \
      const double df_ = (kind == 0) ? t_ : ((kind == 1) ? dq_ : qp_ * t_ + dq_ * ac_);                                \
      const double gv = -(df_ * ax_[kind]);                                                                            \
      if (okl_) AEMIT(ab_[kind] + (((blk) == 0 ? 1 : ((blk) == 1 ? 4 : 7)) + (col)) * n8_ + j08_, gv);                 \
    }                                                                                                                  \
  } while (0)
#if GEL_CA_CACHE
      Bracket ca_br = no_bracket();   // the node's Mach interval, shared by all of its aerodynamic-force evaluations
#define GEL_CA_BRACKET (JAC ? &ca_br : (Bracket*)nullptr)   // a residual-only launch looks up once
#else
#define GEL_CA_BRACKET nullptr
#endif
      // AERO: p and 1/hypot(z Ra, p Rb) are NOT parked (nor is 1/p, below): the position sweeps form them again from the node
      // position (geodetic_p_ih: the statements that produced them, 30 operations per sweep), and their three slots hold the aero
      // rows' centre values cos(alpha), alpha and q instead of ten more registers across the sweeps
      struct ParkSink { lds_double* park; GEL_DEV void put(int i, double v) const { if (!(AERO && (i == PCS_P || i == PCS_IH))) park[(PK_FP0 + i) * 64] = v; } };
      constexpr int PK_ACC = PK_FP0 + PCS_P, PK_AAC = PK_FP0 + PCS_IH, PK_AQC = PK_LV2;   // AERO: cos(alpha_c), alpha_c, q_c
      static_assert(PCS_COUNT == 8 && PK_FP7 == PK_FP0 + 7, "PosCentre's early members fill FP0-7");
      {
        PosPart pp;
        double v[3], w[3], F[3], dir[3];
        const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
        if (JAC) pp = pos_part<true, ParkSink>(r, tb, P.barC20, nullptr, ParkSink{park}, &pt);
        else pp = pos_part(r, tb, P.barC20);
        if (__builtin_amdgcn_ballot_w64(!(pp.wn == 0.0 && pp.we == 0.0)) != 0) {   // tn is still in registers here
          const EarthAngle e0 = earth_angle(tn);
          eh.ch = e0.ch; eh.sh = e0.sh; have_eh = true;
          // AERO: con_aero's Earth angle is formed where the aero rows begin (below), from the node's time waiting in the slot that
          // cos(alpha_c) takes over there -- not here, where its pair would occupy four registers across the centre evaluation
          if (aero_on) PARK_SET(PK_ACC, tn);
        }
        const EarthAngle ea = full_angle(eh);
        wind_eci_or_calm(r, ea, pp.shp, pp.chp, pp.inv_p, pp.wn, pp.we, w);
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = PARK_GET(PK_V0 + c) * P.uv;
        aero_force(r, v, pp.rho, pp.inv_a, ea, w, ph.area, tb, F, GEL_CA_BRACKET);
        // thrust = T * direction is formed where it is used (T from the parked pressure): three registers instead of eight
#define GEL_T (ph.thrust - ph.nozzle * pp.P)
#define GEL_TDC(name) const double name[3] = {GEL_T * dir[0], GEL_T * dir[1], GEL_T * dir[2]}
        {
          const double q[4] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2), PARK_GET(PK_Q3)};
          thrust_dir(q, dir);
        }
        double tm[3];   // (thrust + aerodynamic force) / m: what the mass sweep scales
        {
          GEL_TDC(Tdc);
          accel_parts(Tdc, F, inv_m, pp.g, inv_uv, tm, fc);
        }
        if (rb) {  // velocity defect (:216-289); its D.X row leaves slots LV0-2, which then serve as the tile of the transposed store
#ifndef GEL_RES_XPOSE_VEL_AIR
#define GEL_RES_XPOSE_VEL_AIR 1   // the velocity defect of an aerodynamic phase through the tile as well (with the max-ilp scheduling
                                  // strategy the three values live at the register peak cost no scratch any more): in-process A/B
                                  // -0.35 % at mixed-6x64, level at dense / 12 x 128 (round 5)
#endif
          if (!JAC || GEL_RES_XPOSE_VEL_AIR) {
            double cv[3];
#pragma unroll
            for (int c = 0; c < 3; c++) cv[c] = PARK_GET(PK_LV0 + c) - fc[c] * (tf - to) * hT;
            RSTORE_ROWS(wave_lds + PK_LV0 * 64, 3, rs_v, cv);
          } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const double cv = PARK_GET(PK_LV0 + c) - fc[c] * (tf - to) * hT;
              RSTORE(rs_v + 3 * gn + c, cv);
              GEL_CHK(cv);
            }
          }
        }
        if (JAC) {   // the half-latitude pair and 1/p wait in the slots of the D.X row for the position sweeps
          asm volatile("" ::: "memory");
          PARK_SET(PK_LV0, pp.shp); PARK_SET(PK_LV1, pp.chp);
          if (!AERO) PARK_SET(PK_LV2, pp.inv_p);
        }
        if (JAC && lead && !P.fd_recompute) GEL_MASS_CLOSED(tm);   // first: (T d + F) / m dies here
        // velocity sweeps: only the aerodynamic force changes.  Latency form of a WHOLE evaluation (P.split_vel: the optimiser's
        // callback): the wavefront of position sweep k takes velocity sweep k too -- it has formed the centre's wind, force and
        // thrust anyway -- so the lead wavefront, the longest chain of the launch, is three aerodynamic-force evaluations shorter
        // (same operations on the same operands: the same bits).  Unit-sharded launches keep them with the lead, which owns
        // their slots (gel_unit_owner).
        if (JAC && ph.air_fd && (SPLIT && P.split_vel ? !lead : lead)) {
          double f[3];
          {
            const double djj = PARK_GET(PK_DJJ);
#pragma unroll
            for (int k = 0; k < 3; k++) {
              if (SPLIT && P.split_vel && k != part - 1) continue;   // wave-uniform
              double vp[3], Fp[3];
#pragma unroll
              for (int c = 0; c < 3; c++) vp[c] = ((k == c) ? (PARK_GET(PK_V0 + c) + dx) : PARK_GET(PK_V0 + c)) * P.uv;
              const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
              aero_force(r, vp, pp.rho, pp.inv_a, ea, w, ph.area, tb, Fp, GEL_CA_BRACKET);
              GEL_TDC(Tdc);
              accel(Tdc, Fp, inv_m, pp.g, inv_uv, f);
              // submat_vel[3j+c, 3(j+1)+k] = D[j][j+1]*(c==k) + rh_vel   (con_dynamics.py:341-343,415-416)
#pragma unroll
              for (int c = 0; c < 3; c++) EMIT(ph.s_vv + 3 * k + c, ((c == k) ? djj : 0.0) + FDQ(f[c], fc[c]));
            }
          }
        }
        if (JAC && lead) {
          double f[3];
          // quaternion sweeps: only the thrust direction changes; mass sweep: only the division by mass
          if (!P.fd_recompute) {
            GEL_QUAT_CLOSED(GEL_T);
          } else if (P.fd_recompute) {
            const double q[4] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2), PARK_GET(PK_Q3)};
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
              double qp[4];
#pragma unroll
              for (int c = 0; c < 4; c++) qp[c] = (k == c) ? (q[c] + dx) : q[c];
              double dp[3];
              thrust_dir(qp, dp);
              const double Td[3] = {GEL_T * dp[0], GEL_T * dp[1], GEL_T * dp[2]};
              accel(Td, F, inv_m, pp.g, inv_uv, f);
#pragma unroll
              for (int c = 0; c < 3; c++) EMIT(ph.s_vq + 3 * k + c, FDQ(f[c], fc[c]));
            }
            GEL_TDC(Tdc);
            accel(Tdc, F, frcp((me + dx) * P.um), pp.g, inv_uv, f);
#pragma unroll
            for (int c = 0; c < 3; c++) EMIT(kSlotVM + c, FDQ(f[c], fc[c]));
          }
          // t0 / tf columns (con_dynamics.py:452-480): their two sweeps move only the Earth angle -- and the RHS does not depend on
          // it: the rotation by omega t is applied and undone (src/Coordinate.cpp:41-59), and the NED axes at an inertial position
          // do not move with t, so f_p = f_c up to rounding and the reference's quotient -(f_p (tf_p - to_p) - f_c (tf - to))/dx ut/2
          // is +-f_c ut/2 plus the noise of f_p - f_c (measured against the reference's own values: <= 2e-6 on entries of 109,
          // 1e-8 typically).  The entries are therefore written in that closed form, which is also what the reference itself
          // uses for the phases without aerodynamics (:478-480); GEL_FLAG_FD_RECOMPUTE keeps the two sweeps (below).
          if (!ph.t_fd) {
            const double vt[3] = {fc[0] * hT, fc[1] * hT, fc[2] * hT};  // t0 column; tf = its negative
            EMIT_GROUP(3, ph.s_vt, vt, CG_VT, 0, true, true);
          }
        }
        if (aero_on) {
          // ---- aero rows, centre (lib/con_aero.py:39-87,127-139): the air-relative velocity with con_aero's Earth angle, alpha, q;
          //      then the sweeps whose inputs are at hand here -- quaternion (only the body axis changes) and velocity (only the
          //      air-relative velocity) -- and the t columns (exact zeros: aero_body, gel_kernels.hip)
          asm volatile("" ::: "memory");
          const double ra[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
          double va[3], wa[3], a0[3];
#pragma unroll
          for (int c = 0; c < 3; c++) va[c] = PARK_GET(PK_V0 + c) * P.uv;
          // the wind in ECI with con_aero's Earth angle (aero_body: wind_eci_or_calm at the centre).  What the rotation needs of the
          // centre's position part is taken from where it lies by now -- the half-latitude pair from the park, 1/p and the wind
          // components formed again (geodetic_p_ih, pos_centre_tail: bit-identical) -- instead of five values kept in registers
          // across the centre evaluation
          if (have_eh) {   // the centre's wind is not calm in some lane (wave-uniform): the node's time in seconds (lib/con_aero.py:45)
            const EarthAngle e1 = earth_angle(PARK_GET(PK_ACC) * P.ut);
            aeh.ch = e1.ch; aeh.sh = e1.sh; have_aeh = true;
          }
          if (have_aeh) {
            PosCentre pcx;
            double wn_, we_, p_, ip_, ih_;
            pos_centre_tail(pt, pp.rho, pp.P, tb, pcx, wn_, we_);
            geodetic_p_ih(ra[0], ra[1], ra[2], p_, ip_, ih_);
            wind_eci(ra, full_angle(GEL_AEH), PARK_GET(PK_LV0), PARK_GET(PK_LV1), ip_, wn_, we_, wa);
          } else {
            wa[0] = 0.0; wa[1] = 0.0; wa[2] = 0.0;
          }
          const double nv2 = aero_vair2(ra, va, wa, a0);
          const double ind = frsqrt(fmax(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2], 1.0e-300));
          const double cc = a_need_alpha ? aero_cos(a0, nv2, dir, ind) : 0.0;
          const double alpha_c = a_need_alpha ? aero_acos(cc, nv2) : 0.0;
          const double qdyn_c = 0.5 * pp.rho * nv2;                       // 0.5 rho |v_air|^2 (wrapper_utils.hpp:163-175)
          double sc, isc;
          fsqrt_rsqrt(fmax((1.0 - cc) * (1.0 + cc), 1.0e-300), sc, isc);  // sin(alpha_c) and its reciprocal
          const bool centre_ok = (cc <= 1.0) && (nv2 >= 1.0e-12) && (sc > 1.0e-6);
          GEL_AERO_FETCH();
          const double ail_[3] = {load_const(&aph->il[0]), load_const(&aph->il[1]), load_const(&aph->il[2])};
#pragma unroll
          for (int kind = 0; kind < 3; kind++) {
            const double il = ail_[kind];
            const double cv = 1.0 - ((kind == 0) ? alpha_c : (kind == 1) ? qdyn_c : qdyn_c * alpha_c) * il;
            AEMIT(ab_[kind] + j08_, cv);
            GEL_CHK(cv);
            // (the t0 / tf columns: exact zeros, not stored -- gel_aero_record_map names them -1)
          }
          if (a_need_alpha && !GEL_X_NOQUAT) {
            const double q[4] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2), PARK_GET(PK_Q3)};
#pragma unroll
            for (int c = 0; c < 4; c++) {
              double qp[4], dp[3];
#pragma unroll
              for (int d = 0; d < 4; d++) qp[d] = (d == c) ? q[d] + dx : q[d];
              thrust_dir(qp, dp);
              const double indp = frsqrt(fmax(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2], 1.0e-300));
              GEL_AERO_ENTRY(2, c, a0, nv2, dp, indp, pp.rho, true, cc, alpha_c, qdyn_c, isc, centre_ok);
              __builtin_amdgcn_sched_barrier(0);   // one sweep at a time: interleaved, the seven sweeps of this block cost six spilled registers
            }
          }
#pragma unroll
          for (int c = 0; c < (GEL_X_NOVEL ? 0 : 3); c++) {
            double vp[3], a[3];
#pragma unroll
            for (int d = 0; d < 3; d++) vp[d] = ((d == c) ? PARK_GET(PK_V0 + d) + dx : PARK_GET(PK_V0 + d)) * P.uv;
            const double nv2p = aero_vair2(ra, vp, wa, a);
            GEL_AERO_ENTRY(1, c, a, nv2p, dir, ind, pp.rho, true, cc, alpha_c, qdyn_c, isc, centre_ok);
            __builtin_amdgcn_sched_barrier(0);
          }
          // for the position sweeps: in the slots of p, 1/hypot and 1/p (1/sin(alpha_c) is formed again there)
          PARK_SET(PK_ACC, cc); PARK_SET(PK_AAC, centre_ok ? alpha_c : -alpha_c); PARK_SET(PK_AQC, qdyn_c);
        }
        if (JAC) {   // quaternion and D[j][j+1] are no longer needed: their slots take f_c and the thrust direction
          asm volatile("" ::: "memory");
          PARK_SET(PK_Q0, fc[0]); PARK_SET(PK_Q1, fc[1]); PARK_SET(PK_Q2, fc[2]); PARK_SET(PK_Q3, dir[0]); PARK_SET(PK_DJJ, dir[1]);
          dir2 = dir[2];
          cen_rho = pp.rho; cen_P = pp.P; cen_inv_a = pp.inv_a;
        }
#undef GEL_TDC
#undef GEL_T
      }
      // position sweeps (lib/con_dynamics.py:381-400); SPLIT: this wavefront's one
      if (JAC && (!SPLIT || part)) {
        const int k0 = SPLIT ? part - 1 : 0, k1 = SPLIT ? part : 3;
        // the tail of a sweep, given the position part at the perturbed point: the perturbed RHS value f_
#define GEL_POS_SWEEP_F(rp, pq, f_)                                                                           \
  do {                                                                                                        \
    double wq_[3], Fp_[3];                                                                                    \
    GEL_NEED_EARTH_ANGLE((pq).wn, (pq).we);                                                                   \
    const EarthAngle ea = full_angle(eh);                                                                     \
    wind_eci_or_calm(rp, ea, (pq).shp, (pq).chp, (pq).inv_p, (pq).wn, (pq).we, wq_);                          \
    const double vq_[3] = {PARK_GET(PK_V0) * P.uv, PARK_GET(PK_V1) * P.uv, PARK_GET(PK_V2) * P.uv};           \
    aero_force(rp, vq_, (pq).rho, (pq).inv_a, ea, wq_, ph.area, tb, Fp_, GEL_CA_BRACKET);                     \
    const double Tp_ = ph.thrust - ph.nozzle * (pq).P;                                                        \
    const double Td_[3] = {Tp_ * PARK_GET(PK_Q3), Tp_ * PARK_GET(PK_DJJ), Tp_ * dir2};                        \
    accel(Td_, Fp_, inv_m, (pq).g, inv_uv, f_);                                                               \
  } while (0)
#define GEL_POS_SWEEP_EMIT(kk, f_, okl_)                                                                      \
  do {                                                                                                        \
    const double vp_[3] = {FDQ((f_)[0], PARK_GET(PK_Q0)), FDQ((f_)[1], PARK_GET(PK_Q1)), FDQ((f_)[2], PARK_GET(PK_Q2))}; \
    EMIT_GROUP(3, kSlotVP + 3 * (kk), vp_, CG_VP, kk, false, okl_);                                           \
  } while (0)
        // AERO: the aero rows' position entries of sweep kk at the perturbed point (aero_body's GEL_AERO_POS_TAIL): wind into ECI with
        // con_aero's Earth angle, air-relative velocity, the body axis from the park (1 / |d| formed again: not carried)
#define GEL_AERO_POS(kk, rp, pq, okl_)                                                                        \
  do {                                                                                                        \
    double wq_[3], a_[3];                                                                                     \
    GEL_AERO_NEED_EA((pq).wn, (pq).we);                                                                       \
    wind_eci_or_calm(rp, full_angle(GEL_AEH), (pq).shp, (pq).chp, (pq).inv_p, (pq).wn, (pq).we, wq_);             \
    const double vq_[3] = {PARK_GET(PK_V0) * P.uv, PARK_GET(PK_V1) * P.uv, PARK_GET(PK_V2) * P.uv};           \
    const double nv2_ = aero_vair2(rp, vq_, wq_, a_);                                                         \
    const double dd_[3] = {PARK_GET(PK_Q3), PARK_GET(PK_DJJ), dir2};                                          \
    const double ind_ = frsqrt(fmax(dd_[0] * dd_[0] + dd_[1] * dd_[1] + dd_[2] * dd_[2], 1.0e-300));          \
    const double cc_ = PARK_GET(PK_ACC);                                                                      \
    double sc_, isc_;                                                                                         \
    fsqrt_rsqrt(fmax((1.0 - cc_) * (1.0 + cc_), 1.0e-300), sc_, isc_);                                        \
    const double acs_ = PARK_GET(PK_AAC);   /* alpha_c, negated where the difference form does not apply at the centre */ \
    GEL_AERO_FETCH();                                                                                         \
    GEL_AERO_ENTRY(0, kk, a_, nv2_, dd_, ind_, (pq).rho, okl_, cc_, fabs(acs_), PARK_GET(PK_AQC), isc_, !__builtin_signbit(acs_)); \
  } while (0)
        // PosCentre and the centre's position part as far as pos_delta() reads them, from the park
#define GEL_LOAD_POS_CENTRE(pc, pcv)                                                                                   \
  PosCentre pc;                                                                                                        \
  pc.ihy = PARK_GET(PK_FP0 + PCS_IHY);                                                                                 \
  pc.sl = PARK_GET(PK_FP0 + PCS_SL); pc.cl = PARK_GET(PK_FP0 + PCS_CL); pc.icl = PARK_GET(PK_FP0 + PCS_ICL);           \
  pc.N = PARK_GET(PK_FP0 + PCS_N); pc.G = PARK_GET(PK_FP0 + PCS_G);                                                    \
  PosPart pcv;                                                                                                         \
  pcv.rho = cen_rho; pcv.P = cen_P; pcv.inv_a = cen_inv_a;                                                             \
  pcv.shp = PARK_GET(PK_LV0); pcv.chp = PARK_GET(PK_LV1);                                                              \
  if (AERO) {                                                                                                          \
    geodetic_p_ih(fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up), pc.p, pcv.inv_p, pc.ih); \
  } else {                                                                                                             \
    pc.p = PARK_GET(PK_FP0 + PCS_P); pc.ih = PARK_GET(PK_FP0 + PCS_IH); pcv.inv_p = PARK_GET(PK_LV2);                  \
  }                                                                                                                    \
  pos_centre_tail(pt, cen_rho, cen_P, tb, pc, pcv.wn, pcv.we)
        unsigned todo = 0;   // sweeps with a lane the difference form does not cover (wave-uniform)
        unsigned long long coo_okm = 0;   // COO-direct output: the verdict of the difference form, kept for the recomputing pass (the LDS tile
                                          // of the first pass's stores has overwritten the parked PosCentre by then)
        if (P.fd_recompute) {
          todo = ((1u << k1) - 1u) & ~((1u << k0) - 1u);
        } else {
#pragma unroll
          for (int k = k0; k < k1; k++) {
            asm volatile("" ::: "memory");   // PosCentre is read from the park inside every trip
            GEL_LOAD_POS_CENTRE(pc, pcv);
            const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
            double rp[3];
#pragma unroll
            for (int c = 0; c < 3; c++) rp[c] = (k == c) ? (re[c] + dx) * P.up : r[c];
            const double dlt = (k == 0) ? rp[0] - r[0] : ((k == 1) ? rp[1] - r[1] : rp[2] - r[2]);   // exact
            PosPart pq;
            const bool ok = pos_delta(r, k, dlt, pcv, pc, tb, pq);
            if (coo) coo_okm = __builtin_amdgcn_ballot_w64(ok);   // the latency form runs ONE sweep per wavefront
            if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
              todo |= 1u << k;
              if (__builtin_amdgcn_ballot_w64(ok) == 0) continue;
            }
            gravity_eci(rp, P.barC20, pq.g);
            double f[3];
            GEL_POS_SWEEP_F(rp, pq, f);
            // in a wavefront with a lane the difference form does not cover, the covered lanes write now and the others after
            // the recomputation below: every lane writes once, and a covered lane always writes the value of the difference
            // form -- a lane's entries do not depend on which other nodes (or, with two vectors per wavefront, which other
            // decision vector) share its wavefront
            GEL_POS_SWEEP_EMIT(k, f, ok);
            if (aero_on && !GEL_X_NOPOS) GEL_AERO_POS(k, rp, pq, ok);
          }
        }
        if (todo) {
          // the chain once more on the perturbed position, for the lanes the difference form does not cover (all lanes of a
          // problem created with GEL_FLAG_FD_RECOMPUTE)
#pragma unroll 1
          for (int k = k0; k < k1; k++) {
            if (!((todo >> k) & 1u)) continue;
            asm volatile("" ::: "memory");
            double rp[3];
#pragma unroll
            for (int c = 0; c < 3; c++) rp[c] = fresh_product((k == c) ? re[c] + dx : re[c], P.up);
            bool ok = false;   // a lane mask on the scalar unit: the verdict of the difference form once more
            if (!P.fd_recompute && coo) {
              ok = ((coo_okm >> lane) & 1ull) != 0;
            } else if (!P.fd_recompute) {
              GEL_LOAD_POS_CENTRE(pc, pcv);
              const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
              const double dlt = (k == 0) ? rp[0] - r[0] : ((k == 1) ? rp[1] - r[1] : rp[2] - r[2]);
              PosPart pd;
              ok = pos_delta(r, k, dlt, pcv, pc, tb, pd);
            }
            asm volatile("" ::: "memory");
            PosPart pq = pos_part(rp, tb, P.barC20);
            double f[3];
            GEL_POS_SWEEP_F(rp, pq, f);
            GEL_POS_SWEEP_EMIT(k, f, !ok);
            if (aero_on && !GEL_X_NOPOS) GEL_AERO_POS(k, rp, pq, !ok);
          }
        }
#undef GEL_AERO_POS
#undef GEL_LOAD_POS_CENTRE
#undef GEL_NEED_EARTH_ANGLE
#undef GEL_AERO_NEED_EA
#undef GEL_AERO_ENTRY
#undef GEL_AERO_FETCH
#undef GEL_POS_SWEEP_F
#undef GEL_POS_SWEEP_EMIT
      }
      if (JAC && lead && ph.t_fd) {
        // GEL_FLAG_FD_RECOMPUTE: the t0 / tf sweeps (con_dynamics.py:452-480); only the Earth angle changes.  This form is for
        // audits, not for speed: the position part of the centre is formed once more (bit-identical) and parked, so that only
        // the node position lives in registers across the sincos chains of earth_angle(); quaternion and node abscissa are
        // fetched again (a load behind the stores: it waits for them).
        asm volatile("" ::: "memory");
        const double fcc[3] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2)};
        int jl2 = jc;
        if (AERO) asm volatile("" : "+v"(jl2));
        const double tau2 = P.tau[ph.toff + jl2];
        {
          const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
          const PosPart p2 = pos_part(r, tb, P.barC20);
          const double T2 = ph.thrust - ph.nozzle * p2.P;
          PARK_SET(PK_FP0, p2.shp); PARK_SET(PK_FP1, p2.chp); PARK_SET(PK_FP2, p2.inv_p); PARK_SET(PK_FP3, p2.wn); PARK_SET(PK_FP4, p2.we);
          PARK_SET(PK_FP5, p2.rho); PARK_SET(PK_FP6, p2.inv_a);
          PARK_SET(PK_LV0, p2.g[0]); PARK_SET(PK_LV1, p2.g[1]); PARK_SET(PK_LV2, p2.g[2]);
          PARK_SET(PK_Q0, T2 * PARK_GET(PK_Q3)); PARK_SET(PK_Q1, T2 * PARK_GET(PK_DJJ)); PARK_SET(PK_Q2, T2 * dir2);
        }
#pragma unroll 1
        for (int k = 0; k < 2; k++) {
          asm volatile("" ::: "memory");  // re-read the park inside every trip
          const double to_p = (k == 0) ? to + dx : to;
          const double tf_p = (k == 1) ? tf + dx : tf;
          const double tnp = tau2 * (tf_p - to_p) / 2 + (tf_p + to_p) / 2;
          const EarthAngle eq = earth_angle(tnp);
          double wq[3], Fp[3], f[3];
          const double rq[3] = {re[0] * P.up, re[1] * P.up, re[2] * P.up};
          wind_eci_or_calm(rq, eq, PARK_GET(PK_FP0), PARK_GET(PK_FP1), PARK_GET(PK_FP2), PARK_GET(PK_FP3), PARK_GET(PK_FP4), wq);
          const double vq[3] = {PARK_GET(PK_V0) * P.uv, PARK_GET(PK_V1) * P.uv, PARK_GET(PK_V2) * P.uv};
          aero_force(rq, vq, PARK_GET(PK_FP5), PARK_GET(PK_FP6), eq, wq, ph.area, tb, Fp, GEL_CA_BRACKET);
          const double Tq[3] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2)};
          const double gq[3] = {PARK_GET(PK_LV0), PARK_GET(PK_LV1), PARK_GET(PK_LV2)};
          accel(Tq, Fp, inv_m, gq, inv_uv, f);
          // -(f_p*(tf_p - to_p) - f_c*(tf - to))/dx*unit_t/2   (con_dynamics.py:463-477)
#pragma unroll
          for (int c = 0; c < 3; c++)
            EMIT(ph.s_vt + 3 * k + c, (fcc[c] * (tf - to) - f[c] * (tf_p - to_p)) * fdt);
        }
      }
    } else {
      // NoAir (reference_area == 0): thrust + gravity only (src/pybind_dynamics.cpp:73-92)
      const double T = ph.thrust;
      double dir[3];
      {
        const double q[4] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2), PARK_GET(PK_Q3)};
        thrust_dir(q, dir);
      }
      const double Td[3] = {T * dir[0], T * dir[1], T * dir[2]};
      double gc[3];
      {
        const double r[3] = {re[0] * P.up, re[1] * P.up, re[2] * P.up};
        gravity_eci(r, P.barC20, gc);
      }
      double tm[3];
#pragma unroll
      for (int c = 0; c < 3; c++) { tm[c] = Td[c] * inv_m; fc[c] = (tm[c] + gc[c]) * inv_uv; }   // accel_noair(), keeping T d / m
      if (rb) {  // velocity defect (:216-289)
        double cv[3];
#pragma unroll
        for (int c = 0; c < 3; c++) cv[c] = PARK_GET(PK_LV0 + c) - fc[c] * (tf - to) * hT;
        RSTORE_ROWS(wave_lds + PK_LV0 * 64, 3, rs_v, cv);
      }
      if (JAC) {
        double f[3];
        if (!P.fd_recompute) {
          GEL_MASS_CLOSED(tm);
        } else {
          accel_noair(Td, frcp((me + dx) * P.um), gc, inv_uv, f);
#pragma unroll
          for (int c = 0; c < 3; c++) EMIT(kSlotVM + c, FDQ(f[c], fc[c]));
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          double r[3], gp[3];
#pragma unroll
          for (int c = 0; c < 3; c++) r[c] = ((k == c) ? (re[c] + dx) : re[c]) * P.up;
          gravity_eci(r, P.barC20, gp);
          accel_noair(Td, inv_m, gp, inv_uv, f);
          const double vp[3] = {FDQ(f[0], fc[0]), FDQ(f[1], fc[1]), FDQ(f[2], fc[2])};
          EMIT_GROUP(3, kSlotVP + 3 * k, vp, CG_VP, k, false, true);
        }
        if (!P.fd_recompute) {
          GEL_QUAT_CLOSED(T);
        } else {
          const double q[4] = {PARK_GET(PK_Q0), PARK_GET(PK_Q1), PARK_GET(PK_Q2), PARK_GET(PK_Q3)};
#pragma unroll 1
          for (int k = 0; k < 4; k++) {
            double qp[4];
#pragma unroll
            for (int c = 0; c < 4; c++) qp[c] = (k == c) ? (q[c] + dx) : q[c];
            double dp[3];
            thrust_dir(qp, dp);
            const double Tp[3] = {T * dp[0], T * dp[1], T * dp[2]};
            accel_noair(Tp, inv_m, gc, inv_uv, f);
#pragma unroll
            for (int c = 0; c < 3; c++) EMIT(ph.s_vq + 3 * k + c, FDQ(f[c], fc[c]));
          }
        }
        {
          const double vt[3] = {fc[0] * hT, fc[1] * hT, fc[2] * hT};  // t0 column; tf = its negative
          EMIT_GROUP(3, ph.s_vt, vt, CG_VT, 0, true, true);
        }
      }
    }
  }
#undef EMIT_GROUP
#undef GEL_COO_EMIT
#undef GEL_COO_BASE
#undef EMIT
#undef GEL_CA_BRACKET
#undef GEL_QUAT_CLOSED
#undef GEL_MASS_CLOSED
#undef AEMIT
#undef EMIT_AT
#undef RSTORE
#undef RSTORE_ROWS
#undef RSTORE_LINE
#undef xpose
#undef FDQ
#undef GEL_CHK
#undef GEL_UNI
#undef GEL_TAKE_KNOT_TIMES
#undef PARK_GET
#undef PARK_SET
  if (bad) *(volatile int32_t*)P.flag = 1;  // every writer stores the same 1
}

// Last thing a workgroup of a self-signalling launch does (ProblemDev::done_flag): its wavefronts' stores released at system scope,
// itself counted; the last workgroup of the grid resets the counter and hands the launch's sequence number to the host.
__device__ __forceinline__ void signal_done(const ProblemDev& P) {
  if (!P.done_flag) return;                       // wave-uniform (kernel argument)
  __threadfence_system();                         // this wavefront's results (pinned host memory) are visible system-wide ...
  __syncthreads();                                // ... and so are those of the workgroup's other wavefronts
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(P.done_ctr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == P.done_total - 1) {
      __hip_atomic_store(P.done_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch (which the host starts after the flag)
      __hip_atomic_store(P.done_flag, P.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// LONGP = false: the launcher vouches that no phase has kXldsPipeFrom nodes or more, and the instantiation is without the slab
// loop (used by the residual-only cooperative form, which then needs 72 VGPRs instead of 118)
// NTS = false: the Jacobian values by ordinary stores instead of non-temporal ones -- launches whose output fits the Infinity Cache
// and is read again at once (full COO values by update in place at B = 1024: 3.9 -> 6.2 M evals/s; at B = 65536 ordinary stores
// cost 4.4 %)
// AERO = true: the aero path constraints' rows of the aerodynamic phases' nodes ride along (P.aero_ph / aero_out / aero_ld)
template <bool JAC, bool MFMA, bool SPLIT = false, bool PACK = false, bool LONGP = true, bool NTS = true, bool AERO = false>
__global__ __launch_bounds__(kBlock, (!JAC && PACK) ? GEL_MIN_WAVES_PER_SIMD_RES : ((JAC && PACK) ? GEL_MIN_WAVES_PER_SIMD_PACKJAC : GEL_MIN_WAVES_PER_SIMD)) void eval_kernel(ProblemDev P, int B, const double* __restrict__ x,
                                                      double* __restrict__ res, double* __restrict__ jvar) {
  eval_body<JAC, MFMA, SPLIT, PACK, 9, LONGP, NTS, AERO>(P, B, x, res, jvar, blockIdx.x);
  if constexpr (SPLIT) signal_done(P);   // latency form only: the throughput forms never signal
}

}  // namespace gel
