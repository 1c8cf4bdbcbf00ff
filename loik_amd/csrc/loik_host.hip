// loik_host.hip -- host driver + C-ABI (include/loik_amd.h) of the batched LoIK solver for MI355X.
//
// Host-side mirror of the reference's orchestration (paths under /root/reference/):
//   loikb_create          <- IkIdDataTypeOptimizedTpl ctor + FirstOrderLoikOptimizedTpl ctor
//                            (include/loik/loik-loid-data-optimized.hxx:40-104, loik-loid-optimized.hpp:129-162)
//   loikb_solve_init      <- SolveInit          (loik-loid-optimized.hpp:335-361)
//   loikb_solve           <- Solve()            (loik-loid-optimized.hpp:368-455)
//   loikb_solve_full      <- Solve(q,H_ref,...) (loik-loid-optimized.hpp:475-580)
//   loikb_solve_tailored  <- Solve(q,c_id,A,b)  (loik-loid-optimized.hpp:596-695)
// Reset semantics follow IkIdDataTypeOptimizedTpl::Reset / ResetRecursion
// (loik-loid-data-optimized.hxx:114-154) and IkProblemFormulationOptimized (ik-id-description-optimized.hpp).
//
// Device memory: wavefront tiles (see loik_device.hpp).  Set 0 is the "home" set (tile t, lane l holds instance
// 64 t + l); sets 1 and 2 are half-size work sets used by lane compaction.
#include "loik_device.hpp"
#include "loik_tail.hpp"
#include "loik_lean.hpp"
#include "loik_flat.hpp"
#include "loik_flat2.hpp"
#include "loik_passes.hpp"
#ifdef LOIKB_FLAT_SEPARATE_TU
// k_flat2 / k_flat1 are instantiated in loik_flat_kernels.hip (its own code-generation switches: loik_flat_inst.hpp); here they are only launched
#include "loik_flat_inst.hpp"
LOIKB_FLAT2_INSTANCES(LOIKB_FLAT2_DECL)
LOIKB_FLAT1_INSTANCES(LOIKB_FLAT1_DECL)
#ifdef LOIKB_TAIL_PROF
int loikb_flat_prof_read(unsigned long long* out, int which, int reset);   // (loik_flat_kernels.hip: that unit's copy of the phase counters)
#endif
#endif

#include "../../include/loik_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace loikb;

static thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) {                                                                                \
      char _buf[512];                                                                                      \
      snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      g_last_error = _buf;                                                                                 \
      return LOIKB_ERR_HIP;                                                                                \
    }                                                                                                      \
  } while (0)

namespace {

constexpr int ROWMAP_CAP = 8192;

// Developer overrides from the environment, read ONCE in loikb_create (never inside a solve).  Defaults are the measured
// choices; the variables exist for the experiments recorded in DESIGN.md / profiles/ and for tests/test_engines.py.
struct Tuning {
  int team = MAX_TEAM;          // LOIKB_TEAM          wavefronts per tile of k_solve
  int team_max = 1 << 30;       // LOIKB_TEAM_MAX      use the team schedule only up to this many slots
  int tile_pad = -1;            // LOIKB_TILE_PAD      extra pairs per tile (-1: pad to an odd number of 1-KiB pairs)
  bool lean = true;             // LOIKB_LEAN=0        never use k_lean nor k_flat (the engines with precomputed decade slots)
  bool flat = true;             // LOIKB_FLAT=0        never use k_flat (the engine without level loops, loik_flat.hpp)
  bool flat_split = true;       // LOIKB_FLAT_SPLIT=0: k_flat (one joint per lane) also where k_flat2 (two lanes per joint) applies
  int flat_split_wpe = 2;       // LOIKB_FLAT_WPE=3: k_flat2 built for three wavefronts per SIMD
  int flat_win_lo = 0, flat_win_n = 0;   // LOIKB_FLAT_WINDOW=lo,n: (with LOIKB_FLAT_BUILD=1) k_fslots builds decades lo .. lo + n - 1 only, the rest lazily
  int flat_build = 2;           // (default 2 since the end of round 5; 0: the full table always) LOIKB_FLAT_BUILD=1: the lazily populated table on every k_flat2 launch (window: LOIKB_FLAT_WINDOW, else all of it), 2: on time-sliced launches, window from the handle's history; k_flat2 builds a decade slot its table lacks in-wave (flat_build_slot) instead of handing the instance to k_tail
  int fslot_dgrp = 0;           // LOIKB_FSLOT_DGRP=g: k_fslots takes the decades through its two passes g at a time (default: all)
  int flat_slice2 = 0;          // LOIKB_FLAT_SLICE2=q: an instance's later slices (0: as the first)
  int flat_small_batch = 2048;  // LOIKB_FLAT_SMALL_BATCH=n: up to n instances a Solve() on k_flat2 / k_flat1 is the short sequence (fused queue set-up, no order pass)
  int flat_min_batch = 1;       // LOIKB_FLAT_MIN_BATCH=n: k_flat2 / k_flat1 take batches from n instances (round 5: 64; below, k_tail)
  int small_finish = 1;         // LOIKB_SMALL_FINISH=0: the short sequence ends with k_list_unfinished + a copy of the counters + run_main_loop's own event and
                                //   synchronisation, as before (1: k_small_finish writes the counters into the pinned host copy; one synchronisation per Solve())
  int small_slot_event = 0;     // LOIKB_SMALL_SLOT_EVENT=1: an event between k_fslots and the engine in the short sequence too (default: none, loikb_stats::hslots_ms reads 0
                                //   there -- an event costs a lone problem's Solve() 2.5 us of 116: profiles/r06_o_small_finish_ab.jsonl)
  int flat_probe = 0;           // (default 0: measured a wash against the round robin, profiles/r06_a_probe_and_finish_ab.txt; given: wherever LOIKB_FLAT_SLICE / the default slices it) LOIKB_FLAT_PROBE=p: a time-sliced k_flat2 launch without an order becomes TWO launches -- every instance for p iterations at
                                // most, then the survivors to completion, longest predicted first (0: one launch, round robin: round 5's)
  int flat_probe_mark = 32;     // LOIKB_FLAT_PROBE_MARK=k: the probe's first mark (the residual's rate of fall is taken between it and the probe's end)
  int flat_slice = -1;          // LOIKB_FLAT_SLICE=q: k_flat2's / k_flat1's round-robin time slice in iterations (0: never; default: 288 for
                                // launches in arrival order of >= 32 768 instances, see flat_slice_for / run_tail)
  bool flat_zero_state = true;  // LOIKB_FLAT_ZERO_STATE=0: k_flat2 / k_flat1 fetch vis, fis, g, w, z of every instance even straight after a cold reset
  int flat_order_holdoff = 4;   // LOIKB_FLAT_ORDER_HOLDOFF=n: solves in arrival order after an ordered launch that was not shorter (0: never hold off)
  int flat_one_slot = 1;        // LOIKB_FLAT_ONE_SLOT=0: k_flat1 always keeps two decade slots in LDS
  bool flat_order = true;       // LOIKB_FLAT_ORDER=0: the flat engine takes its instances in arrival order even when the handle's previous
                                // solve left an order (longest first, k_order_*: loik_lean.hpp)
  int tail_waves = TAIL_WAVES;  // LOIKB_TAIL_WAVES    wavefronts per k_tail workgroup
  int lean_decades = 10;        // LOIKB_LEAN_DECADES  decades of mu with precomputed H slots ...
  int lean_klo = -2;            // LOIKB_LEAN_KLO      ... starting at mu0 * 10^klo
  int lean_wg_per_cu = 0;       // LOIKB_LEAN_WG_PER_CU (0: what registers and LDS allow)
  int lean_wg_waves = 0;        // LOIKB_LEAN_WG_WAVES wavefronts per k_lean workgroup (0: by plan)
  bool lean_adapt = true;       // LOIKB_LEAN_ADAPT=0  always build the whole configured range of decade slots
  std::vector<int> lean_quanta; // LOIKB_LEAN_QUANTA   host-side rounds of the lean launch (default: none)
  int lean_slice = 0;           // LOIKB_LEAN_SLICE    in-kernel round-robin time slice (0: run to completion)
  double compact_ratio = 0.85;  // LOIKB_COMPACT_RATIO repack k_solve's tiles when at most this share of the slots is live
  bool direct_tail = true;      // LOIKB_NO_DIRECT_TAIL small batches go to the on-chip engines from the first iteration
  int lat_iters = 16;           // LOIKB_LAT_ITERS     iterations per k_solve launch in its latency-bound regime
  int chunks = 0;               // LOIKB_CHUNKS        concurrent chunks (0: by engine plan)
  bool trace = false;           // LOIKB_TRACE         per-launch lines on stderr
  void read_env()
  {
    auto geti = [](const char* n, int& v) { if (const char* e = getenv(n)) v = atoi(e); };
    geti("LOIKB_TEAM", team); team = std::max(1, std::min(MAX_TEAM, team));
    geti("LOIKB_TEAM_MAX", team_max);
    geti("LOIKB_TILE_PAD", tile_pad);
    if (const char* e = getenv("LOIKB_LEAN")) lean = atoi(e) != 0;
    if (const char* e = getenv("LOIKB_FLAT")) flat = atoi(e) != 0;
    if (!lean) flat = false;  // (LOIKB_LEAN=0 asks for the engines without precomputed factors: k_solve / k_tail)
    if (const char* e = getenv("LOIKB_FLAT_SPLIT")) flat_split = atoi(e) != 0;
    if (const char* e = getenv("LOIKB_FLAT_WPE")) flat_split_wpe = atoi(e) == 3 ? 3 : 2;
    if (const char* e = getenv("LOIKB_FLAT_BUILD")) flat_build = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("LOIKB_FLAT_WINDOW")) { if (sscanf(e, "%d,%d", &flat_win_lo, &flat_win_n) != 2) flat_win_n = 0; }
    if (const char* e = getenv("LOIKB_FLAT_SLICE")) flat_slice = std::min(65535, std::max(-1, atoi(e)));
    if (const char* e = getenv("LOIKB_FLAT_SLICE2")) flat_slice2 = std::min(8191, std::max(0, atoi(e)));
    if (const char* e = getenv("LOIKB_FLAT_PROBE")) flat_probe = std::min(8000, std::max(0, atoi(e)));
    if (const char* e = getenv("LOIKB_FLAT_MIN_BATCH")) flat_min_batch = std::max(1, atoi(e));
    if (const char* e = getenv("LOIKB_SMALL_FINISH")) small_finish = atoi(e);
    if (const char* e = getenv("LOIKB_SMALL_SLOT_EVENT")) small_slot_event = atoi(e);
    if (const char* e = getenv("LOIKB_FLAT_SMALL_BATCH")) flat_small_batch = std::max(0, atoi(e));
    if (const char* e = getenv("LOIKB_FLAT_PROBE_MARK")) flat_probe_mark = std::max(1, atoi(e));
    if (const char* e = getenv("LOIKB_FSLOT_DGRP")) fslot_dgrp = std::max(0, atoi(e));
    if (const char* e = getenv("LOIKB_FLAT_ORDER")) flat_order = atoi(e) != 0;
    if (const char* e = getenv("LOIKB_FLAT_ZERO_STATE")) flat_zero_state = atoi(e) != 0;
    if (const char* e = getenv("LOIKB_FLAT_ONE_SLOT")) flat_one_slot = atoi(e);
    if (const char* e = getenv("LOIKB_FLAT_ORDER_HOLDOFF")) flat_order_holdoff = std::max(0, atoi(e));
    geti("LOIKB_TAIL_WAVES", tail_waves); tail_waves = std::max(1, std::min(TAIL_WAVES, tail_waves));
    geti("LOIKB_LEAN_DECADES", lean_decades); lean_decades = std::max(1, std::min(15, lean_decades));   // (15: a parked instance's ring entry says "no slot" with decade code 15 -- ADVICE r05)
    geti("LOIKB_LEAN_KLO", lean_klo);
    geti("LOIKB_LEAN_WG_PER_CU", lean_wg_per_cu);
    geti("LOIKB_LEAN_WG_WAVES", lean_wg_waves);
    if (const char* e = getenv("LOIKB_LEAN_ADAPT")) lean_adapt = atoi(e) != 0;
    if (const char* e = getenv("LOIKB_LEAN_QUANTA"))
      for (const char* p = e; *p;) { lean_quanta.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
    geti("LOIKB_LEAN_SLICE", lean_slice); lean_slice = std::max(0, lean_slice);
    if (const char* e = getenv("LOIKB_COMPACT_RATIO")) compact_ratio = atof(e);
    direct_tail = getenv("LOIKB_NO_DIRECT_TAIL") == nullptr;
    geti("LOIKB_LAT_ITERS", lat_iters);
    geti("LOIKB_CHUNKS", chunks);
    trace = getenv("LOIKB_TRACE") != nullptr;
  }
};

// Which kernels a solve of this handle uses -- decided in ONE place (plan_engines) from (nb, nc, A shared?, children per
// joint, batch, options), whenever one of them changes (create, SolveInit):
//   lean        k_hslots + k_lean take whole batches / the hand-over from k_solve (else k_tail does)
//   tail_max    hand the solve over to the on-chip engine once at most this many instances are live (whole batch: direct)
//   nchunks     concurrent chunks of the k_solve phase (1 when the on-chip engine runs the whole batch in one launch)
struct EnginePlan {
  bool flat = false;               // k_fslots + k_flat take whole batches (when the reference cost of the solve allows: flat_applicable)
  const char* why_not_flat = "";
  int flat_waves_cu = 0;
  bool lean = false;
  const char* why_not_lean = "";
  int lean_waves_cu = 0, lean_wg_waves = TAIL_WAVES;
  int ndec = 10, kexp_lo = -2;
  int tail_max = 32768;
  int nchunks = 1;
  // k_solve (the streaming engine) keeps the leaf->root hand-over of a sweep in LDS slots of the workgroup: one per pending branch.  A very
  // bushy tree (8-10 children at one joint of a 40-joint tree) needs more of them than a CU has LDS.  Such a robot is not refused (round 6):
  // whole batches go to the on-chip engines, which have no such slots (k_flat2 / k_flat1 / k_tail: any number of children), and where those
  // do not apply -- more than 64 joints, or options that ask for k_solve's own behaviour -- to the plain pass-by-pass implementation
  // (k_pass_solve: one instance per thread, state in HBM, no LDS at all), the engine of last resort.
  bool solve_ok = true;
  size_t solve_lds_need = 0;
};

struct loikb_solver_impl {
  Tuning tune;
  EnginePlan plan;
  // model (copied).  nj/nb/parents/jtype/idx_q/jd describe the DEVICE tree, in which every joint has one DoF: a
  // free-flyer / spherical / translation joint of the caller's model is a chain of 6 / 3 / 3 one-DoF joints about the
  // axes of one frame with massless links in between (build_schedule).  nq, nv are the caller's; nv == nb.
  int nj = 0, nb = 0, nq = 0, nv = 0;
  std::vector<int> parents, jtype, idx_q, idx_v;
  std::vector<JointDesc> jd;
  // the caller's model: ext_nj joints; link_of[i] / first_of[i] = device joint that carries body i / that carries
  // M(q) of joint i (the last / first joint of its chain; both i's own image for a 1-DoF joint)
  int ext_nj = 0;
  std::vector<int> link_of, first_of;
  int* d_link_sel = nullptr;  // [ext_nj - 1] link_of[1..] - 1 on the device (getters)
  int* d_first_sel = nullptr;
  // step schedules of the sweeps: [0] one wavefront per tile, [1] a team of wavefronts per tile
  struct TeamSched {
    int nw = 1, T_up = 0, T_down = 0, nslots = 0, nvslots = 0;
    std::vector<StepDesc> up, down;
    std::vector<int> rlist;
    StepDesc *d_up = nullptr, *d_down = nullptr;
    int* d_rlist = nullptr;
  } sched[2];
  // level schedule of the cooperative tail kernel (one joint per lane)
  std::vector<TailTopo> topo;
  std::vector<int> child_list;
  int maxdepth = 0, maxchild = 0;
  int multi_from = 1;  // leaf->root level loop of k_lean: the step from which joints with several children can be final
  TailTopo* d_topo = nullptr;
  int* d_child_list = nullptr;
  // the flat engine's view of the static tree (build_flat_schedule; loik_flat.hpp): one FlatLane per lane of a group
  struct FlatSched {
    bool ok = false;
    const char* why = "";
    int G = 0, nanc = 0, nscan = 0, njmp = 0;
    int fblk = 0;  // scalars per decade slot of an instance: max(8 G, sum of the joints' depths)
    std::vector<FlatLane> lanes;
    FlatLane* d_lanes = nullptr;
  } flat;
  bool ud_stale = false;  // a solve went through the flat engine: the getters of UDinv / pis rebuild them first (k_rebuild_ud)
  // options
  loikb_options opt{};
  int B = 0, nc = 0;
  size_t esz = 8;
  bool f32 = false;
  // problem (uniform part)
  double Href[36]{}, vref[6]{}, Hv[6]{};
  double Hv_inf_norm = 0.0;
  // the same per device joint: [nj][HREF_ROW] = (H_ref_i, H_ref_i v_ref_i), rows of the universe and of massless chain
  // links zero.  Broadcast by SolveInit (UpdateReference), per link after loikb_update_references (UpdateReferences).
  std::vector<double> href_tab;
  double mu_start = 0.0;        // LOIKB_MU_MAXEIGENVALUE: the solve's starting mu (spectral_mu0), else unused
  bool per_link = false;
  void* d_href = nullptr;       // the table in the solve precision
  double* d_href64 = nullptr;   // and in double (== d_href for an fp64 solver): residual-vector getter
  std::vector<float> href_tab32;
  std::vector<int> active_ids;     // [nc_active] = active_task_constraint_ids_
  // `nc` constraint slots exist on the device (eq_c_capacity >= num_eq_c); nc_active = nc_eq_ of them carry a constraint,
  // the others are null constraints (k_edit_constraints) parked on bodies that have none
  int nc_active = 0;
  std::vector<double> A_host;      // [nc][36] the shared A per slot (host mirror: RemoveEqConstraint moves entries down)
  bool have_problem = false;
  bool have_q = false;
  bool a_shared = true, bnd_shared = true;
  bool href_diag = true;  // H_ref diagonal -> k_solve<T, true>
  // device
  int device = 0;
  int ncu = 256;  // compute units of the device
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;  // LOIKB_OPT_OWN_STREAM
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  std::vector<void*> allocs;
  std::mutex alloc_mu;
  JointDesc* d_jd = nullptr;
  int* d_idx_q = nullptr;
  int* d_rowmap = nullptr;             // the row map the next upload / gather kernel reads: a cached one (rowmap_cache) or the scratch
  int* d_rowmap_scratch = nullptr;     // [ROWMAP_CAP]: maps beyond the cache
  struct RowmapEntry { std::vector<int> rm; int* d; };
  std::vector<RowmapEntry> rowmap_cache;   // a handle uses a dozen distinct row maps (lb, ub, A, b, the getters' members ...), again and again: kept on the
                                       //   device -- SolveInit of ONE problem spent more in their uploads and synchronisations than in its kernels
  bool defer_sync = false;             // inside SolveInit / the tailored Solve: the uploads' synchronisations are left to the entry point's own at its end
  bool offer_queue = false;            // the full / tailored Solve: FwdPassInit's closing reset goes straight into the main loop -- reset_home(.., with_queue)
  // ... and, inside such a section, small host inputs do not travel by a copy operation at all: they are put into a pinned, device-visible
  // buffer (h_pin, bump-allocated from pin_off; the section's final synchronisation makes it free again) and the upload kernels read them there
  char* h_pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;
  std::vector<double> uni_host;         // fp64 handles: the host's copy of d_uni -- inside a deferred section upload_uni only writes here, flush_uni sends it once
  bool uni_dirty = false;
  std::vector<int> cslots_on_device;    // the joints' constraint slots as upload_jd last sent them: a SolveInit with the same task links sends nothing
  double* d_q = nullptr;               // [B][nq] configurations resident on the device (outer loop)
  void* d_uni = nullptr;               // A[nc][36], AtA[nc][21], lb[nb], ub[nb] (T)
  void* d_stage = nullptr;             // staging for host<->device copies
  void* d_getscr[2] = {nullptr, nullptr};  // scratch of the getters that rebuild members (His / pis / UDinv): kept, not malloc'ed per call
  // loikb_get_results: the row map of the requested members on the device (rebuilt when the request or the constraint set changes) and a
  // pinned, device-visible host buffer the gather kernel writes into
  struct ResMap { int* d = nullptr; int key[6] = {-1, -1, -1, -1, -1, -1}; int n = 0; int off[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };   // key: mask, nc_active, nb, nl, off_c, crec
  ResMap resmaps[4];                   // (a caller uses one or two masks; replaced round robin)
  int resmap_next = 0;
  double* h_res = nullptr;
  size_t h_res_bytes = 0;
  size_t getscr_bytes[2] = {0, 0};
  size_t stage_bytes = 0;
  // tile layouts: A per instance needs the long constraint record
  Layout L{};
  struct Set {
    char* tiles = nullptr;
    int ntiles = 0;
    int *map = nullptr, *wave_live = nullptr, *wave_off = nullptr;
  };
  Set home;  // tile t, lane l <-> instance 64 t + l
  // The batch is solved as `chunks.size()` independent contiguous ranges of tiles, each driven by its own host
  // thread on its own stream (instances never interact).  While one chunk is in its latency-bound straggler phase
  // (few wavefronts resident) the others keep the machine busy with their bandwidth-bound bulk phase, and the
  // host-side gaps between launches (counter read-back, compaction scan) of one chunk are covered by the others.
  struct Chunk {
    int first_tile = 0, B = 0;           // instances [64 first_tile, 64 first_tile + B)
    Set set[3];                          // [0] view of the home tiles of the range, [1],[2] work sets of the compaction
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_k2 = nullptr, ev_k3 = nullptr;
    unsigned int* d_counters = nullptr;
    unsigned int* h_counters = nullptr;  // pinned
    int* d_slots = nullptr;              // list of the live instances handed to the tail kernel
    int* d_slots2 = nullptr;             // instances that left the lean kernel unfinished (same capacity)
    int* d_order = nullptr;              // the chunk's instances, longest first by the iteration counts of the previous solve
    unsigned int* d_order_bins = nullptr;  // [2 ORDER_BINS] counts / offsets of the counting sort
    int order_n = 0;                     // instances d_order lists (0: none yet)
    unsigned long long order_epoch = 0;  // S->inputs_epoch of the solve whose iteration counts d_order sorts
    int order_holdoff = 0;               // solves to go in arrival order after an order that predicted badly
    int arrival_n = 0;                   // the flat engine's last launch in arrival order: instances, ...
    double arrival_ms = 0.0;             // ... its duration (0: none yet)
    int* d_ring = nullptr;               // work queue of the lean kernel: ring of instance slots (ring_cap, a power of two)
    int ring_cap = 0;
    void* d_hslots = nullptr;            // decade slots of the lean tail kernel (H, Dinv, UDinv per joint and decade)
    size_t hslots_bytes = 0;
    void* d_fslots = nullptr;            // decade slots of the flat engine (W rows, Dinv per joint and decade)
    size_t fslots_bytes = 0;
    void* d_park = nullptr;              // k_flat2<.., SLICED>: where instances whose time slice is used up are parked
    unsigned int* d_fmask = nullptr;     // k_flat2<.., MUR = 2>: per instance, which decades of the slot table are populated
    size_t fmask_n = 0;
    void** d_aux = nullptr;              // k_flat2's cold pointers behind one argument: {TailTopo*, child list, fmask}
    const void* aux_h[4] = {nullptr, nullptr, nullptr, nullptr};   // (what d_aux holds)
    size_t park_bytes = 0;
    std::vector<int> h_wave;             // host scratch for the compaction scan
    loikb_stats stats{};
    bool unfinished_counted = false;     // stats.n_unfinished of the last solve came with the on-chip launch's own counters (small batches)
    bool queue_ready = false;            // list + ring + counters of the whole home set were prepared by this solve's reset launch (reset_home, with_queue)
    bool small_finished = false;         // ... and the launch's last kernel was k_small_finish, waited for in run_tail: nothing of this solve is pending on the stream
    int rc = 0;
    std::string err;
    // [start, end) of every solve / tail launch of the last call, ms since the fork event (HIP events)
    std::vector<std::pair<float, float>> solve_iv, tail_iv;
  };
  std::vector<Chunk> chunks;
  hipEvent_t ev_fork = nullptr;
  loikb_stats stats{};
  // pass-level debug path (loik_passes.hpp): the data object of the reference, field by field, per instance
  bool pass_active = false;
  bool zero_state = false;   // the solve in progress began with a reset that zeroed vis, fis, g, w, z of every instance
  // Anything that changes what a solve computes (SolveInit, a tailored solve's q / constraint, UpdateEqConstraint, UpdateReferences,
  // Add / RemoveEqConstraint, integrate, the setters) bumps this.  A handle's later solves are taken longest first by the previous
  // solve's iteration counts only when those counts were taken on THESE inputs (a cold Solve() of the same problem is
  // deterministic: the counts are exact) -- or when the caller says that consecutive problems resemble each other
  // (LOIKB_OPT_ORDER_FROM_PREVIOUS: a tracking planner), and then under the watch of the timing comparison below.
  unsigned long long inputs_epoch = 1;
  bool href_known = false;   // S->Href holds the reference weight of the problem being set up (loikb_solve_init, before the plan is made)
  bool tab_bcast = false;    // the links' reference table on the device holds ONE (H_ref, v_ref) pair for all links: tab_H / tab_v
  double tab_H[36] = {0}, tab_v[6] = {0};
  int log_truncated = 0;     // instances of the last logged solve whose SolverInfo lists end early (see run_logged)
  PassLayout PL{};
  double* d_pass = nullptr;
  int* d_pass_cslot = nullptr;
  // decades of mu the lean engine's instances actually visited in the solves of this handle so far (absolute exponents): the
  // next solve builds slots for [seen_lo - 1, seen_hi + 1] only (within the configured range); an escape widens it again
  int seen_lo = 1 << 20, seen_hi = -(1 << 20);
  unsigned long long end_hist[32] = {0}, end_hist_n = 0;   // decades (kexp + 16) the instances of the last flat solve ended in
  unsigned long long solve_serial = 0, end_hist_epoch = 0; // (run_main_loop_t counts the solves; the solve end_hist was taken in)
  // SolverInfo lists of a handle created with logging = 1 (k_pass_solve)
  double* d_log = nullptr;
  int* d_log_rows = nullptr;
  int log_rows_cap = 0;
  bool have_log = false;
};
using Chunk = loikb_solver_impl::Chunk;

// `st`: the stream that will use the buffer first (the zero fill is ordered on it); chunk threads allocate their
// work sets concurrently, hence the lock around the bookkeeping
int alloc_dev(loikb_solver_impl* S, void** out, size_t bytes, hipStream_t st = nullptr, bool use_st = false)
{
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes ? bytes : 16));
  HIPCHK(hipMemsetAsync(p, 0, bytes ? bytes : 16, use_st ? st : S->stream));
  {
    std::lock_guard<std::mutex> lock(S->alloc_mu);
    S->allocs.push_back(p);
  }
  *out = p;
  return LOIKB_OK;
}

int ensure_stage(loikb_solver_impl* S, size_t bytes)
{
  if (bytes <= S->stage_bytes) return LOIKB_OK;
  if (S->d_stage) HIPCHK(hipFree(S->d_stage));
  S->d_stage = nullptr;
  S->stage_bytes = 0;
  HIPCHK(hipMalloc(&S->d_stage, bytes));
  S->stage_bytes = bytes;
  return LOIKB_OK;
}

int ensure_getscr(loikb_solver_impl* S, int k, size_t bytes)
{
  if (bytes <= S->getscr_bytes[k]) return LOIKB_OK;
  if (S->d_getscr[k]) HIPCHK(hipFree(S->d_getscr[k]));
  S->d_getscr[k] = nullptr;
  S->getscr_bytes[k] = 0;
  HIPCHK(hipMalloc(&S->d_getscr[k], bytes));
  S->getscr_bytes[k] = bytes;
  return LOIKB_OK;
}

// the getters' scratch goes back when a solve-side buffer needs the room (decade slots, park records): ADVICE r04
static size_t release_getscr(loikb_solver_impl* S)
{
  size_t freed = 0;
  for (int k = 0; k < 2; ++k) {
    if (S->d_getscr[k]) { (void)hipFree(S->d_getscr[k]); S->d_getscr[k] = nullptr; freed += S->getscr_bytes[k]; S->getscr_bytes[k] = 0; }
  }
  (void)hipGetLastError();
  return freed;
}

inline dim3 grid1(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }
inline size_t pair_b(const loikb_solver_impl* S) { return (size_t)WAVE * 2 * S->esz; }
inline size_t tile_bytes(const loikb_solver_impl* S) { return (size_t)S->L.tile_pairs * pair_b(S); }

int alloc_set(loikb_solver_impl* S, loikb_solver_impl::Set& W, int ntiles, hipStream_t st = nullptr, bool use_st = false)
{
  if (W.tiles) return LOIKB_OK;
  int rc;
  void* tmp = nullptr;
  if ((rc = alloc_dev(S, &tmp, tile_bytes(S) * ntiles, st, use_st))) return rc;
  W.tiles = (char*)tmp;
  W.ntiles = ntiles;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * (size_t)ntiles * WAVE, st, use_st))) return rc;
  W.map = (int*)tmp;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * ((size_t)ntiles + 1), st, use_st))) return rc;
  W.wave_live = (int*)tmp;
  if ((rc = alloc_dev(S, &tmp, sizeof(int) * ((size_t)ntiles + 1), st, use_st))) return rc;
  W.wave_off = (int*)tmp;
  return LOIKB_OK;
}


// List-schedule the joints of the two sweep directions onto `nw` wavefronts (see StepDesc in loik_device.hpp).
// Leaf->root: a joint is ready once all its children were handled at earlier steps; root->leaf: once its parent
// was.  A wavefront prefers to continue along its chain (hand-over in registers), free wavefronts start the ready
// joint with the longest remaining path.  nw = 1 degenerates to a depth-first walk.
void build_team_schedule(const std::vector<int>& parents, int nw, loikb_solver_impl::TeamSched& out)
{
  const int nj = (int)parents.size(), nb = nj - 1;
  std::vector<std::vector<int>> children(nj);
  std::vector<int> depth(nj, 0), height(nj, 1);
  for (int i = 1; i < nj; ++i) { children[parents[i]].push_back(i); depth[i] = depth[parents[i]] + 1; }
  for (int i = nj - 1; i >= 1; --i)
    if (parents[i] > 0 && height[i] + 1 > height[parents[i]]) height[parents[i]] = height[i] + 1;
  out.nw = nw;
  std::vector<int> step_of(nj, -1), wave_of(nj, -1);

  // ---- leaf -> root
  {
    std::vector<std::vector<int>> rows(nw);  // rows[w][t] = joint
    std::vector<int> pending(nj, 0), last(nw, 0);
    for (int i = 1; i < nj; ++i) pending[i] = (int)children[i].size();
    std::vector<char> ready(nj, 0), sched(nj, 0);
    for (int i = 1; i < nj; ++i) ready[i] = pending[i] == 0;
    int ndone = 0, t = 0;
    while (ndone < nb) {
      std::vector<int> pick(nw, 0);
      for (int w = 0; w < nw; ++w) {
        const int p = last[w] ? parents[last[w]] : 0;
        if (p > 0 && ready[p] && !sched[p]) { pick[w] = p; sched[p] = 1; }
      }
      for (int w = 0; w < nw; ++w) {
        if (pick[w]) continue;
        int best = 0;
        for (int i = nj - 1; i >= 1; --i)  // ties: larger index first (the reference's visiting order)
          if (ready[i] && !sched[i] && (best == 0 || depth[i] > depth[best])) best = i;
        if (best) { pick[w] = best; sched[best] = 1; }
      }
      for (int w = 0; w < nw; ++w) {
        rows[w].push_back(pick[w]);
        if (pick[w]) { step_of[pick[w]] = t; wave_of[pick[w]] = w; last[w] = pick[w]; ++ndone; }
      }
      for (int w = 0; w < nw; ++w)  // children handled at step t make their parent ready from step t+1 on
        if (pick[w] && parents[pick[w]] > 0 && --pending[parents[pick[w]]] == 0) ready[parents[pick[w]]] = 1;
      ++t;
    }
    out.T_up = t;
    out.up.assign((size_t)nw * t, StepDesc{});
    out.rlist.clear();
    // hand-over kind of every non-root edge
    std::vector<int> out_reg(nj, 0), slot_of(nj, -1);
    for (int w = 0; w < nw; ++w) {
      int prev = 0;
      for (int tt = 0; tt < t; ++tt) {
        const int j = rows[w][tt];
        if (!j) continue;
        if (prev && parents[prev] == j) out_reg[prev] = 1;
        prev = j;
      }
    }
    // edge slots: interval colouring, a slot is free again at the step after its parent consumed it
    std::vector<int> order;
    for (int i = 1; i < nj; ++i) if (parents[i] > 0 && !out_reg[i]) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return step_of[a] != step_of[b] ? step_of[a] < step_of[b] : a < b; });
    std::vector<int> free_at;  // per slot: first step at which it may be written again
    for (int i : order) {
      int s = -1;
      for (int k = 0; k < (int)free_at.size(); ++k) if (free_at[k] <= step_of[i]) { s = k; break; }
      if (s < 0) { s = (int)free_at.size(); free_at.push_back(0); }
      free_at[s] = step_of[parents[i]] + 1;
      slot_of[i] = s;
    }
    out.nslots = (int)free_at.size();
    for (int w = 0; w < nw; ++w)
      for (int tt = 0; tt < t; ++tt) {
        const int j = rows[w][tt];
        StepDesc& sd = out.up[(size_t)w * t + tt];
        sd.joint = j;
        if (!j) continue;
        sd.rstart = (int)out.rlist.size();
        for (int k = (int)children[j].size() - 1; k >= 0; --k) {
          const int c = children[j][k];
          if (out_reg[c]) sd.flags |= SF_IN_REG;
          else out.rlist.push_back(slot_of[c]);
        }
        sd.nread = (int)out.rlist.size() - sd.rstart;
        if (parents[j] > 0) {
          if (out_reg[j]) sd.flags |= SF_OUT_REG;
          else { sd.flags |= SF_OUT_LDS; sd.wslot = slot_of[j]; }
        }
      }
    if (out.rlist.empty()) out.rlist.push_back(0);
  }

  // ---- root -> leaf
  {
    std::vector<std::vector<int>> rows(nw);
    std::vector<int> last(nw, 0);
    std::vector<char> ready(nj, 0), sched(nj, 0);
    for (int i = 1; i < nj; ++i) ready[i] = parents[i] == 0;
    int ndone = 0, t = 0;
    while (ndone < nb) {
      std::vector<int> pick(nw, 0);
      for (int w = 0; w < nw; ++w) {
        int best = 0;
        if (last[w])
          for (int c : children[last[w]])
            if (ready[c] && !sched[c] && (best == 0 || height[c] > height[best])) best = c;
        if (best) { pick[w] = best; sched[best] = 1; }
      }
      for (int w = 0; w < nw; ++w) {
        if (pick[w]) continue;
        int best = 0;
        for (int i = 1; i < nj; ++i)
          if (ready[i] && !sched[i] && (best == 0 || height[i] > height[best])) best = i;
        if (best) { pick[w] = best; sched[best] = 1; }
      }
      for (int w = 0; w < nw; ++w) {
        rows[w].push_back(pick[w]);
        if (pick[w]) { last[w] = pick[w]; ++ndone; }
      }
      for (int w = 0; w < nw; ++w)
        if (pick[w]) for (int c : children[pick[w]]) ready[c] = 1;
      ++t;
    }
    out.T_down = t;
    out.down.assign((size_t)nw * t, StepDesc{});
    std::vector<int> dstep(nj, -1), vreg(nj, 0), last_read(nj, -1), vslot(nj, -1);
    for (int w = 0; w < nw; ++w) {
      int prev = 0;
      for (int tt = 0; tt < t; ++tt) {
        const int j = rows[w][tt];
        if (!j) continue;
        dstep[j] = tt;
        if (prev && parents[j] == prev) vreg[j] = 1;
        prev = j;
      }
    }
    // v hand-over slots: a joint whose velocity is needed by a child that does not get it in registers writes it
    // to a slot; the slot is free again at the step after its last reader
    for (int i = 1; i < nj; ++i)
      if (parents[i] > 0 && !vreg[i] && dstep[i] > last_read[parents[i]]) last_read[parents[i]] = dstep[i];
    std::vector<int> order;
    for (int i = 1; i < nj; ++i) if (last_read[i] >= 0) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return dstep[a] != dstep[b] ? dstep[a] < dstep[b] : a < b; });
    std::vector<int> free_at;
    for (int i : order) {
      int s = -1;
      for (int k = 0; k < (int)free_at.size(); ++k) if (free_at[k] <= dstep[i]) { s = k; break; }
      if (s < 0) { s = (int)free_at.size(); free_at.push_back(0); }
      free_at[s] = last_read[i] + 1;
      vslot[i] = s;
    }
    out.nvslots = (int)free_at.size();
    for (int w = 0; w < nw; ++w)
      for (int tt = 0; tt < t; ++tt) {
        const int j = rows[w][tt];
        StepDesc& sd = out.down[(size_t)w * t + tt];
        sd.joint = j;
        if (!j) continue;
        if (vreg[j]) sd.flags |= SF_VPAR_REG;
        else if (parents[j] > 0) sd.rstart = vslot[parents[j]];
        if (vslot[j] >= 0) { sd.flags |= SF_OUT_LDS; sd.wslot = vslot[j]; }
      }
  }
}

// build the uniform per-joint schedule from the Pinocchio-style model
int build_schedule(loikb_solver_impl* S, const loikb_model_desc* m)
{
  const int enj = m->njoints;
  if (enj < 2) { g_last_error = "model: no joints"; return LOIKB_ERR_MODEL; }
  // JointModelComposite: the sub-joints of joint i (validated below)
  auto comp_n = [&](int i) { return (m->jtype[i] == LOIKB_J_COMPOSITE && m->comp_count) ? m->comp_count[i] : 0; };
  auto jt_nq = [](int jt) {
    return jt == LOIKB_J_FREEFLYER ? 7 : (jt == LOIKB_J_SPHERICAL || jt == LOIKB_J_PLANAR) ? 4
           : (jt == LOIKB_J_TRANSLATION || jt == LOIKB_J_SPHERICAL_ZYX) ? 3 : ((jt >= LOIKB_J_RUBX && jt <= LOIKB_J_RUBZ) || jt == LOIKB_J_RUBU) ? 2 : 1;
  };
  auto jt_nv = [](int jt) {
    return jt == LOIKB_J_FREEFLYER ? 6
           : (jt == LOIKB_J_SPHERICAL || jt == LOIKB_J_TRANSLATION || jt == LOIKB_J_SPHERICAL_ZYX || jt == LOIKB_J_PLANAR) ? 3 : 1;
  };
  // ---- the caller's model: Pinocchio numbering (parents[i] < i, idx_q / idx_v cumulative in joint order)
  {
    int iq = 0, iv = 0;
    for (int i = 1; i < enj; ++i) {
      const int p = m->parents[i], jt = m->jtype[i];
      if (p < 0 || p >= i) { g_last_error = "model: parents[i] must be < i"; return LOIKB_ERR_MODEL; }
      if (jt < LOIKB_J_RX || jt > LOIKB_J_HU) {
        g_last_error = "model: unsupported joint type (supported: 1-DoF joints incl. unbounded revolute, free-flyer, spherical, "
                       "spherical ZYX, translation, planar, helical, composites of those -- a universal joint is the composite "
                       "of its two revolute joints; not: mimic)";
        return LOIKB_ERR_MODEL;
      }
      if (jt >= LOIKB_J_HX && jt <= LOIKB_J_HU && !m->pitch) { g_last_error = "model: a helical joint needs loikb_model_desc.pitch"; return LOIKB_ERR_MODEL; }
      if (m->idx_q[i] != iq || m->idx_v[i] != iv) {
        g_last_error = "model: idx_q/idx_v must be cumulative in joint order (Pinocchio's layout)";
        return LOIKB_ERR_MODEL;
      }
      if (jt == LOIKB_J_COMPOSITE) {
        if (!m->comp_first || !m->comp_count || !m->comp_jtype || !m->comp_axis || !m->comp_placement || m->comp_count[i] < 1 ||
            m->comp_count[i] > 6 || m->comp_first[i] < 0) {
          g_last_error = "model: a composite joint needs comp_first / comp_count (1..6) / comp_jtype / comp_axis / comp_placement";
          return LOIKB_ERR_MODEL;
        }
        int cnv = 0;
        for (int k = 0; k < m->comp_count[i]; ++k) {
          const int st = m->comp_jtype[m->comp_first[i] + k];
          if (st < LOIKB_J_RX || st > LOIKB_J_HU || st == LOIKB_J_COMPOSITE) {
            g_last_error = "model: a sub-joint of a composite joint must be one of the supported joint types other than a composite";
            return LOIKB_ERR_MODEL;
          }
          if (st >= LOIKB_J_HX && !m->comp_pitch) { g_last_error = "model: a helical sub-joint needs loikb_model_desc.comp_pitch"; return LOIKB_ERR_MODEL; }
          iq += jt_nq(st); iv += jt_nv(st); cnv += jt_nv(st);
        }
        if (cnv > 6) { g_last_error = "model: a composite joint has at most 6 degrees of freedom here"; return LOIKB_ERR_MODEL; }
        continue;
      }
      iq += jt_nq(jt); iv += jt_nv(jt);
    }
    if (m->nq != iq || m->nv != iv) { g_last_error = "model: nq/nv do not match the joint types"; return LOIKB_ERR_MODEL; }
    // (any numbering with parents[i] < i is accepted, as upstream: depth-first like pinocchio::urdf::buildModel, breadth-first
    //  or mixed like a model assembled with addJoint -- the sweep schedules, the level loops and the child lists are built
    //  from `parents` alone)
  }
  // ---- the device tree.  A multi-DoF joint whose S selects columns of I6 (free-flyer: all six, spherical: the angular
  // three, translation: the linear three) becomes a chain of 1-DoF joints about those axes of ONE frame: the first
  // chain joint carries placement * M(q), the others are the identity, and the links between them are massless (no
  // rho I + H_ref, no reference term, not counted in the norms over links).  Eliminating nu_k one coordinate at a time
  // along the chain is the block elimination upstream's calc_aba does with its nv x nv Dinv (hxx:60-63) -- the same
  // Schur complement, so all iterates agree up to rounding (tests/test_multidof.py proves it on the CPU oracle).
  // DoF k of the model is device joint k + 1, so nu / z / w / lb / ub keep the caller's order.
  S->ext_nj = enj;
  S->nq = m->nq; S->nv = m->nv;
  S->link_of.assign(enj, 0);
  S->first_of.assign(enj, 0);
  S->parents.assign(1, 0);
  S->jtype.assign(1, LOIKB_J_NONE);
  S->idx_q.assign(1, 0);
  S->jd.assign(1, JointDesc{});
  for (int i = 1; i < enj; ++i) {
    // the ELEMENTARY joints of joint i: itself, or the sub-joints of a JointModelComposite -- literally the chain of its
    // sub-joints, each with its own placement and coordinates (a multi-DoF sub-joint expands into its own chain below); the
    // bodies between them do not exist (massless), the very last link carries the joint's body
    const int ncomp = comp_n(i);
    const int nelem = ncomp ? ncomp : 1;
    const int par = S->link_of[m->parents[i]];
    S->first_of[i] = (int)S->parents.size();
    int elem_q = m->idx_q[i];  // configuration offset of the next elementary joint
    for (int e = 0; e < nelem; ++e) {
      const int ce = ncomp ? m->comp_first[i] + e : -1;  // entry of the comp_* arrays
      const int jt = ncomp ? m->comp_jtype[ce] : m->jtype[i];
      const double* jaxis = ncomp ? m->comp_axis + 3 * ce : m->axis + 3 * i;
      const int n = jt_nv(jt);
      // placement of the elementary joint's frame seen from the link before it: jointPlacements[i] (* comp_placement[first])
      // for the first one, comp_placement[e] for the later sub-joints of a composite
      double P0[12];
      if (!ncomp) for (int c = 0; c < 12; ++c) P0[c] = m->placement[12 * i + c];
      else {
        const double* Pc = m->comp_placement + 12 * ce;
        if (e == 0) {
          const double* Pj = m->placement + 12 * i;
          for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) P0[3 * r + c] = Pj[3 * r] * Pc[c] + Pj[3 * r + 1] * Pc[3 + c] + Pj[3 * r + 2] * Pc[6 + c];
            P0[9 + r] = Pj[9 + r] + Pj[3 * r] * Pc[9] + Pj[3 * r + 1] * Pc[10] + Pj[3 * r + 2] * Pc[11];
          }
        } else for (int c = 0; c < 12; ++c) P0[c] = Pc[c];
      }
      for (int k = 0; k < n; ++k) {
        JointDesc d{};
        double ax[3] = {0, 0, 0};
        int rot = ROT_NONE, flags = 0, sub = jt;
        if (jt == LOIKB_J_SPHERICAL_ZYX) {
          // R = Rz(q0) Ry(q1) Rx(q2), nu = the three angle rates: literally a chain of three revolute joints about z, y, x
          // with their own coordinates (S(q) of JointModelSphericalZYX::calc is this chain's Jacobian) and massless links
          sub = k == 0 ? LOIKB_J_RZ : k == 1 ? LOIKB_J_RY : LOIKB_J_RX;
        } else if (jt == LOIKB_J_PLANAR) {
          sub = k == 0 ? LOIKB_J_PX : k == 1 ? LOIKB_J_PY : LOIKB_J_RZ;  // ConstraintPlanar: vx, vy, wz of ONE frame
        } else if (jt >= LOIKB_J_RUBX && jt <= LOIKB_J_RUBZ) {
          sub = LOIKB_J_RX + (jt - LOIKB_J_RUBX);
          flags |= JF_CS_DIRECT;
        } else if (jt == LOIKB_J_RUBU) {  // JointModelRevoluteUnboundedUnaligned: a revolute joint about `axis` whose q IS (cos, sin)
          sub = LOIKB_J_RU;
          flags |= JF_CS_DIRECT;
        } else if (jt >= LOIKB_J_HX && jt <= LOIKB_J_HU) {  // JointModelHelical*: the revolute joint about the axis + JF_HELICAL, pitch
          sub = jt == LOIKB_J_HU ? LOIKB_J_RU : LOIKB_J_RX + (jt - LOIKB_J_HX);
          flags |= JF_HELICAL;
          d.pitch = ncomp ? m->comp_pitch[ce] : m->pitch[i];
        } else if (n > 1) {
          // chain joint k: prismatic along / revolute about axis (k mod 3) of the joint frame
          const bool angular = jt == LOIKB_J_SPHERICAL || (jt == LOIKB_J_FREEFLYER && k >= 3);
          sub = (angular ? LOIKB_J_RX : LOIKB_J_PX) + k % 3;
        }
        switch (sub) {
        case LOIKB_J_RX: ax[0] = 1; rot = ROT_X; flags |= JF_REVOLUTE; break;
        case LOIKB_J_RY: ax[1] = 1; rot = ROT_Y; flags |= JF_REVOLUTE; break;
        case LOIKB_J_RZ: ax[2] = 1; rot = ROT_Z; flags |= JF_REVOLUTE; break;
        case LOIKB_J_PX: ax[0] = 1; break;
        case LOIKB_J_PY: ax[1] = 1; break;
        case LOIKB_J_PZ: ax[2] = 1; break;
        case LOIKB_J_RU: for (int c = 0; c < 3; ++c) ax[c] = jaxis[c]; rot = ROT_U; flags |= JF_REVOLUTE; break;
        case LOIKB_J_PU: for (int c = 0; c < 3; ++c) ax[c] = jaxis[c]; break;
        }
        for (int c = 0; c < 9; ++c) d.Rp[c] = k > 0 ? (c % 4 == 0 ? 1.0 : 0.0) : P0[c];
        for (int c = 0; c < 3; ++c) d.tp[c] = k > 0 ? 0.0 : P0[9 + c];
        if (n > 1 && jt != LOIKB_J_SPHERICAL_ZYX) {  // (a ZYX chain consists of ordinary revolute joints: each reads its own angle)
          if (k == 0) rot = jt == LOIKB_J_FREEFLYER ? ROT_FREE : jt == LOIKB_J_SPHERICAL ? ROT_SPH
                            : jt == LOIKB_J_PLANAR ? ROT_PLANAR : ROT_TRANS;
          else flags |= JF_NOQ;
        }
        if (!(e == nelem - 1 && k == n - 1)) flags |= JF_MASSLESS;
        for (int c = 0; c < 3; ++c) d.axis[c] = ax[c];
        d.parent = (e == 0 && k == 0) ? par : (int)S->parents.size() - 1;
        if (d.parent == 0) flags |= JF_PARENT_ROOT;
        d.flags = flags;
        d.cslot = -1;
        d.rot = rot;
        S->parents.push_back(d.parent);
        S->jtype.push_back(sub);
        S->idx_q.push_back(jt == LOIKB_J_SPHERICAL_ZYX ? elem_q + k : (k == 0 ? elem_q : 0));
        S->jd.push_back(d);
      }
      elem_q += jt_nq(jt);
    }
    S->link_of[i] = (int)S->parents.size() - 1;
  }
  const int nj = (int)S->parents.size();
  S->nj = nj; S->nb = nj - 1;
  if (S->nb != S->nv) { g_last_error = "internal: device tree size != nv"; return LOIKB_ERR_MODEL; }
  S->idx_v.resize(nj);
  for (int i = 0; i < nj; ++i) S->idx_v[i] = i > 0 ? i - 1 : 0;
  build_team_schedule(S->parents, 1, S->sched[0]);
  build_team_schedule(S->parents, S->tune.team, S->sched[1]);
  // depth / children of every joint for the level-synchronous tail kernel
  S->topo.assign(nj, TailTopo{});
  S->child_list.clear();
  S->maxdepth = 0; S->maxchild = 0;
  for (int i = 1; i < nj; ++i) {
    S->topo[i].depth = S->parents[i] == 0 ? 1 : S->topo[S->parents[i]].depth + 1;
    if (S->topo[i].depth > S->maxdepth) S->maxdepth = S->topo[i].depth;
  }
  for (int i = 1; i < nj; ++i) {
    S->topo[i].child_start = (int)S->child_list.size();
    for (int c = nj - 1; c > i; --c)  // decreasing joint index = the order of the reference's leaf->root sweep
      if (S->parents[c] == i) S->child_list.push_back(c - 1);  // lane of the child
    S->topo[i].nchild = (int)S->child_list.size() - S->topo[i].child_start;
    if (S->topo[i].nchild > S->maxchild) S->maxchild = S->topo[i].nchild;
  }
  {
    // height of a joint = steps of the leaf->root level loop after which its message is final (a leaf: 1)
    std::vector<int> height(nj, 1);
    for (int i = nj - 1; i >= 1; --i)
      if (S->parents[i] > 0 && height[i] + 1 > height[S->parents[i]]) height[S->parents[i]] = height[i] + 1;
    S->multi_from = 1 << 30;
    for (int i = 1; i < nj; ++i)
      if (S->topo[i].nchild > 1 && height[i] < S->multi_from) S->multi_from = height[i];
  }
  return LOIKB_OK;
}


// The flat engine's view of the static tree (loik_flat.hpp): lane j of a group carries device joint j + 1, the joints must be
// numbered depth-first (every subtree a contiguous range of lanes -- Pinocchio's numbering of a parsed model), at most
// FLAT_MAXA strict ancestors per joint.  Per lane: depth, subtree size, the ancestors at distance 1, 2, 4, ... (path sums by
// pointer jumping), the ancestors by depth (the W rows), and its share of the products W_{a,d} tau_d: row a of that sum runs
// over the descendants of a, <= FLAT_RED terms go to lane a itself, the rest in chunks of FLAT_RED to lanes that have no
// row of their own (leaves, unused lanes of the group), which publish partial sums.  `parents` is the device tree.
void build_flat_schedule(const std::vector<int>& parents, loikb_solver_impl::FlatSched& out)
{
  out = loikb_solver_impl::FlatSched{};
  const int nj = (int)parents.size(), nb = nj - 1;
  if (nb > WAVE) { out.why = "more joints than lanes of a wavefront"; return; }
  int G = 8;
  while (G < nb) G <<= 1;
  std::vector<int> depth(nj, 0), size(nj, 1);
  int maxdepth = 0, maxsize = 0;
  for (int i = 1; i < nj; ++i) { depth[i] = parents[i] == 0 ? 1 : depth[parents[i]] + 1; maxdepth = std::max(maxdepth, depth[i]); }
  for (int i = nj - 1; i >= 1; --i) if (parents[i] > 0) size[parents[i]] += size[i];
  for (int i = 1; i < nj; ++i) {
    maxsize = std::max(maxsize, size[i]);
    for (int a = parents[i]; a > 0; a = parents[a])
      if (!(a < i && i < a + size[a])) { out.why = "joints not numbered depth-first (a subtree is not a contiguous range)"; return; }
  }
  // (contiguity of every subtree: the joints in [a, a + size_a) must all descend from a -- counted above from the other side:
  //  every descendant lies inside; there are size_a - 1 of them and the range holds size_a - 1 joints)
  if (maxdepth - 1 > FLAT_MAXA) { out.why = "tree deeper than the flat engine's ancestor table"; return; }
  int njmp = 0;
  while ((1 << njmp) < maxdepth) ++njmp;
  if (njmp > FLAT_JMP) { out.why = "tree deeper than the flat engine's jump table"; return; }
  int nscan = 0;
  while ((1 << nscan) <= maxsize) ++nscan;
  out.lanes.assign(G, FlatLane{});
  for (int l = 0; l < G; ++l) {
    FlatLane& F = out.lanes[l];
    for (int& x : F.jmp) x = -1;
    for (int& x : F.anc) x = -1;
    for (int& x : F.red) x = -1;
    for (int& x : F.part) x = -1;
    if (l >= nb) continue;
    const int i = l + 1;
    F.depth = depth[i]; F.size = size[i];
    int a = i;
    for (int dist = 0, r = 0; a > 0; a = parents[a], ++dist) {
      if (dist > 0) F.anc[depth[a] - 1] = a - 1;
      if (dist == (1 << r) && r < FLAT_JMP) F.jmp[r++] = a - 1;
    }
  }
  // the products' rows: lane a sums the entries (depth_a - 1, lane') of its descendants lane' = a+1 .. a+size_a-1
  std::vector<int> free_lanes;
  for (int l = G - 1; l >= 0; --l)
    if (l >= nb || size[l + 1] == 1) free_lanes.push_back(l);  // (taken from the back of the vector: low lanes first)
  for (int l = 0; l < nb; ++l) {
    const int m = size[l + 1] - 1, k = depth[l + 1] - 1;
    int nparts = 0;
    for (int c0 = 0; c0 < m; c0 += FLAT_RED) {
      int owner = l;
      if (c0 > 0) {
        if (free_lanes.empty() || nparts >= FLAT_PART) { out.why = "a joint with too many descendants for the flat engine's reduction schedule"; out.lanes.clear(); return; }
        owner = free_lanes.back(); free_lanes.pop_back();
        out.lanes[owner].helper = 1;
        out.lanes[l].part[nparts++] = owner;
      }
      for (int t = 0; t < FLAT_RED && c0 + t < m; ++t) out.lanes[owner].red[t] = k * G + (l + 1 + c0 + t);
    }
  }
  out.G = G; out.nanc = std::max(1, maxdepth - 1); out.nscan = nscan; out.njmp = njmp;
  // packed decade slots (fslotW_at, loik_flat.hpp): joint j's column starts at the sum of the depths before it
  int col = 0;
  for (int l = 0; l < G; ++l) {
    out.lanes[l].helper |= col << 8;
    col += out.lanes[l].depth;
  }
  out.fblk = (std::max(FSLOT_ROWS * G, col) + 1) & ~1;
  out.ok = true;
}

template <typename T>
Bufs<T> make_bufs(loikb_solver_impl* S, Chunk* C, int k)
{
  Bufs<T> Bf{};
  Bf.tiles = C->set[k].tiles;
  Bf.uni = (const T*)S->d_uni;
  Bf.counters = C->d_counters;
  Bf.wave_live = C->set[k].wave_live;
  Bf.log = S->d_log; Bf.log_rows = S->d_log_rows; Bf.log_cap = S->log_rows_cap; Bf.log_B = S->B;
  return Bf;
}

// ---- LOIKB_MU_MAXEIGENVALUE (extension: declared upstream, task-solver-base.hpp:13-18, and never implemented there --
// loik-loid-optimized.hxx:635-637 throws).  Defined as a spectral initialisation of the penalty followed by DEFAULT's decade
// steps: mu starts at the geometric mean of the extreme eigenvalues of the links' cost blocks rho I + sym(H_ref,i) over all links
// that carry a cost, snapped to a quarter decade (10^(k/4)), clipped to [1e-6, 1e6]; the constructor's mu is not used.  The
// reference's fixture (H_ref = I, rho = 1e-5) starts at mu = 1 instead of 1e-2.  Every engine runs it: for the kernels it is the
// DEFAULT rule with another mu0.  (The CPU checker of the tests implements the same definition on its own.)
static void jacobi6(double* a, double* ev)  // eigenvalues of a symmetric 6x6 (overwritten), cyclic Jacobi
{
  for (int sweep = 0; sweep < 50; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) off += a[6 * p + q] * a[6 * p + q];
    if (off < 1e-300) break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        const double apq = a[6 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[6 * q + q] - a[6 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 6; ++k) {
          const double akp = a[6 * k + p], akq = a[6 * k + q];
          a[6 * k + p] = c * akp - sn * akq;
          a[6 * k + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k) {
          const double apk = a[6 * p + k], aqk = a[6 * q + k];
          a[6 * p + k] = c * apk - sn * aqk;
          a[6 * q + k] = sn * apk + c * aqk;
        }
      }
  }
  for (int k = 0; k < 6; ++k) ev[k] = a[7 * k];
}

static double spectral_mu0(const loikb_solver_impl* S)
{
  double lo = 0.0, hi = 0.0;
  bool any = false;
  for (int i = 1; i < S->nj; ++i) {
    if (S->jd[i].flags & JF_MASSLESS) continue;  // (the intermediate links of a multi-DoF joint's chain carry no cost)
    const double* H = S->href_tab.data() + (size_t)i * HREF_ROW;
    double m[36], ev[6];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) m[6 * r + c] = 0.5 * (H[6 * r + c] + H[6 * c + r]) + (r == c ? S->opt.rho : 0.0);
    jacobi6(m, ev);
    for (int k = 0; k < 6; ++k) {
      if (!any || ev[k] < lo) lo = ev[k];
      if (!any || ev[k] > hi) hi = ev[k];
      any = true;
    }
  }
  if (!any) return S->opt.mu;
  lo = std::max(lo, S->opt.rho);
  hi = std::max(hi, lo);
  if (!(lo > 0.0)) return S->opt.mu;
  const long q = std::lround(4.0 * std::log10(std::sqrt(lo * hi)));
  return std::min(1e6, std::max(1e-6, std::pow(10.0, (double)q / 4.0)));
}

// mu a solve starts from (and the decades of the on-chip engines are counted from)
static double solve_mu0(const loikb_solver_impl* S)
{
  return S->opt.mu_update_strat == LOIKB_MU_MAXEIGENVALUE && S->mu_start > 0.0 ? S->mu_start : S->opt.mu;
}

template <typename T>
Params<T> make_params(loikb_solver_impl* S)
{
  Params<T> P{};
  for (int k = 0; k < 36; ++k) P.Href[k] = (T)S->Href[k];
  for (int k = 0; k < 6; ++k) P.Hv[k] = (T)S->Hv[k];
  P.Hv_inf_norm = (T)S->Hv_inf_norm;
  P.href_tab = S->per_link ? (const T*)S->d_href : nullptr;
  P.rho = (T)S->opt.rho; P.mu0 = (T)solve_mu0(S); P.mu_scale = (T)S->opt.mu_equality_scale_factor;
  P.tol_abs = (T)S->opt.tol_abs; P.tol_rel = (T)S->opt.tol_rel; P.tol_primal_inf = (T)S->opt.tol_primal_inf;
  P.tol_tail_solve = (T)S->opt.tol_tail_solve;
  P.max_iter = S->opt.max_iter;
  int mode = 0;
  if (S->opt.flags & LOIKB_OPT_FIXED_ITERS) mode |= MODE_FIXED_ITERS;
  if (!(S->opt.flags & LOIKB_OPT_NO_H_CACHE)) mode |= MODE_CACHE_H;
  if (S->a_shared) mode |= MODE_A_SHARED;
  if (S->bnd_shared) mode |= MODE_BND_SHARED;
  if (S->opt.mu_update_strat == LOIKB_MU_OSQP) mode |= MODE_MU_OSQP;
  P.mode = mode;
  P.B = S->B;
  P.max_launch_iters = S->opt.max_iter + 1;
  P.L = S->L;
  return P;
}

// one k_reset launch over the home set
int reset_home(loikb_solver_impl* S, int what, bool with_queue = false)
{
  const dim3 grid(S->home.ntiles), block(WAVE);
  // (a solve starts with RS_SOLVER: does it start from vis = fis = g = w = z = 0?  MODE_ZERO_STATE of the flat engine's launch)
  if (what & RS_SOLVER) S->zero_state = (what & (RS_DATA_COLD | RS_RECURSION)) != 0;
  // (with_queue: the caller goes straight into the main loop.  A small fp64 handle's on-chip launch takes its list, ring and counters from this
  //  launch -- run_tail's short sequence, which checks and clears queue_ready; a solve that goes elsewhere prepares its own as ever)
  if (with_queue && !S->f32 && S->chunks.size() == 1 && S->tune.small_finish && S->tune.flat_split && S->B <= S->tune.flat_small_batch &&
      S->chunks[0].d_ring != nullptr && S->chunks[0].d_slots != nullptr && S->chunks[0].d_counters != nullptr) {
    loikb_solver_impl::Chunk& C = S->chunks[0];
    const int qblocks = (std::max(C.ring_cap, NCOUNTERS) + WAVE - 1) / WAVE;
    hipLaunchKernelGGL(k_reset_and_queue<double>, dim3(S->home.ntiles + qblocks), block, 0, S->stream, S->home.tiles, S->L, what, (double)solve_mu0(S),
                       S->home.ntiles, C.d_ring, C.ring_cap, C.d_slots, S->B, C.d_counters, NCOUNTERS);
    HIPCHK(hipGetLastError());
    C.queue_ready = true;
    return LOIKB_OK;
  }
  if (S->f32) hipLaunchKernelGGL(k_reset<float>, grid, block, 0, S->stream, S->home.tiles, S->L, what, (float)solve_mu0(S));
  else hipLaunchKernelGGL(k_reset<double>, grid, block, 0, S->stream, S->home.tiles, S->L, what, (double)solve_mu0(S));
  HIPCHK(hipGetLastError());
  return LOIKB_OK;
}

// make `src` (host or device, `bytes`) readable by a kernel: returns a device pointer
int to_device(loikb_solver_impl* S, const void* src, size_t bytes, bool src_device, const void** out)
{
  if (src_device) { *out = src; return LOIKB_OK; }
  if (S->defer_sync && bytes <= ((size_t)64 << 10)) {
    // (one problem per call: q, lb, ub, A, b are a few hundred bytes each -- five copy operations cost SolveInit 40 us of its 120)
    if (S->h_pin == nullptr) {
      if (hipHostMalloc((void**)&S->h_pin, (size_t)512 << 10) == hipSuccess) S->pin_cap = (size_t)512 << 10;
      else { S->h_pin = nullptr; (void)hipGetLastError(); }
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (S->h_pin && S->pin_off + need <= S->pin_cap) {
      memcpy(S->h_pin + S->pin_off, src, bytes);
      *out = S->h_pin + S->pin_off;
      S->pin_off += need;
      return LOIKB_OK;
    }
  }
  int rc = ensure_stage(S, bytes);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(S->d_stage, src, bytes, hipMemcpyHostToDevice, S->stream));
  *out = S->d_stage;
  return LOIKB_OK;
}

int set_rowmap(loikb_solver_impl* S, const std::vector<int>& rm)
{
  if ((int)rm.size() > ROWMAP_CAP) { g_last_error = "row map too large"; return LOIKB_ERR_ARG; }
  for (const loikb_solver_impl::RowmapEntry& e : S->rowmap_cache)
    if (e.rm == rm) { S->d_rowmap = e.d; return LOIKB_OK; }   // (on the device already: no copy, no synchronisation)
  if (S->d_rowmap_scratch == nullptr) S->d_rowmap_scratch = S->d_rowmap;
  if (S->rowmap_cache.size() < 64 && !rm.empty()) {
    void* p = nullptr;
    int rc = alloc_dev(S, &p, sizeof(int) * rm.size());
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(p, rm.data(), sizeof(int) * rm.size(), hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));   // (once per distinct map: rm is a local of the caller)
    S->rowmap_cache.push_back(loikb_solver_impl::RowmapEntry{rm, (int*)p});
    S->d_rowmap = (int*)p;
    return LOIKB_OK;
  }
  S->d_rowmap = S->d_rowmap_scratch;
  // rm is a local of the caller: the copy must complete before it goes out of scope
  HIPCHK(hipMemcpyAsync(S->d_rowmap, rm.data(), sizeof(int) * rm.size(), hipMemcpyHostToDevice, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

// instance-major [B][n] doubles (or one shared [n] replicated to every instance) -> tile elements
int upload_rows(loikb_solver_impl* S, const double* src, const std::vector<int>& rm, bool src_device, bool shared)
{
  const int n = (int)rm.size();
  const void* dsrc = nullptr;
  int rc;
  if ((rc = set_rowmap(S, rm))) return rc;
  if ((rc = to_device(S, src, sizeof(double) * (shared ? (size_t)n : (size_t)S->B * n), src_device && !shared, &dsrc))) return rc;
  if (S->f32)
    hipLaunchKernelGGL(k_upload_rows<float>, grid1(S->B), dim3(256), 0, S->stream, (const double*)dsrc, n, (int)shared,
                       S->d_rowmap, S->L, S->B, S->home.tiles);
  else
    hipLaunchKernelGGL(k_upload_rows<double>, grid1(S->B), dim3(256), 0, S->stream, (const double*)dsrc, n, (int)shared,
                       S->d_rowmap, S->L, S->B, S->home.tiles);
  HIPCHK(hipGetLastError());
  // (the staging buffer is reused by the next upload -- in stream order; the caller's array must outlive the copy: the entry point's own
  //  synchronisation when it defers this one)
  if (!S->defer_sync) HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

// uniform inputs (shared A/AtA, shared bounds) as T
int upload_uni(loikb_solver_impl* S, int offset, const double* src, int n)
{
  if (S->f32) {
    std::vector<float> tmp(n);
    for (int k = 0; k < n; ++k) tmp[k] = (float)src[k];
    HIPCHK(hipMemcpyAsync((float*)S->d_uni + offset, tmp.data(), sizeof(float) * n, hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
  } else {
    const size_t total = (size_t)(S->nc > 0 ? S->nc : 1) * 57 + 2 * (size_t)S->nb;
    if (S->uni_host.size() != total) S->uni_host.assign(total, 0.0);   // (d_uni starts as zeros too)
    memcpy(S->uni_host.data() + offset, src, sizeof(double) * n);
    if (S->defer_sync) { S->uni_dirty = true; return LOIKB_OK; }       // (src may be a local of the caller: nothing asynchronous reads it)
    HIPCHK(hipMemcpyAsync((double*)S->d_uni + offset, src, sizeof(double) * n, hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
  }
  return LOIKB_OK;
}

// what upload_uni left on the host inside a deferred section: ONE copy of the whole (small) buffer, before the first kernel that reads it
int flush_uni(loikb_solver_impl* S)
{
  if (!S->uni_dirty) return LOIKB_OK;
  S->uni_dirty = false;
  HIPCHK(hipMemcpyAsync(S->d_uni, S->uni_host.data(), sizeof(double) * S->uni_host.size(), hipMemcpyHostToDevice, S->stream));   // (a member: it outlives the copy)
  return LOIKB_OK;
}

int upload_jd(loikb_solver_impl* S)
{
  HIPCHK(hipMemcpyAsync(S->d_jd, S->jd.data(), sizeof(JointDesc) * S->nj, hipMemcpyHostToDevice, S->stream));
  // the step schedules carry a copy of each joint's descriptor (cslot changes with the constraint set)
  for (auto& sc : S->sched) {
    for (StepDesc& sd : sc.up) if (sd.joint) sd.d = S->jd[sd.joint];
    for (StepDesc& sd : sc.down) if (sd.joint) sd.d = S->jd[sd.joint];
    HIPCHK(hipMemcpyAsync(sc.d_up, sc.up.data(), sizeof(StepDesc) * sc.up.size(), hipMemcpyHostToDevice, S->stream));
    HIPCHK(hipMemcpyAsync(sc.d_down, sc.down.data(), sizeof(StepDesc) * sc.down.size(), hipMemcpyHostToDevice, S->stream));
  }
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

// FwdPassInit(q), loik-loid-optimized.hxx:253-283.  q == nullptr: the configurations already resident on the device
int fwd_pass_init(loikb_solver_impl* S, const double* q, int in_flags)
{
  ++S->inputs_epoch;
  if (q) {
    const bool shared = in_flags & LOIKB_Q_SHARED;
    const bool dev = (in_flags & LOIKB_IN_DEVICE) && !shared;
    const void* dq = nullptr;
    int rc = to_device(S, q, sizeof(double) * (shared ? (size_t)S->nq : (size_t)S->B * S->nq), dev, &dq);
    if (rc) return rc;
    const long long elems = (long long)S->B * std::max(S->nq, S->nb);
    if (elems <= (1 << 16)) {
      // a small batch: the resident copy and FwdPassInit's pairs from ONE launch with a thread per coordinate / joint (k_set_q_fk_small)
      const dim3 g((unsigned)((elems + 255) / 256));
      if (S->f32) hipLaunchKernelGGL(k_set_q_fk_small<float>, g, dim3(256), 0, S->stream, S->d_q, (const double*)dq, (int)shared, S->nq, S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles);
      else hipLaunchKernelGGL(k_set_q_fk_small<double>, g, dim3(256), 0, S->stream, S->d_q, (const double*)dq, (int)shared, S->nq, S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles);
      HIPCHK(hipGetLastError());
      if (!dev && !S->defer_sync) HIPCHK(hipStreamSynchronize(S->stream));
      S->have_q = true;
      return reset_home(S, RS_HCACHE | (S->opt.warm_start ? 0 : RS_Y), S->offer_queue);   // (as below)
    }
    // the resident copy is what the outer loop advances (loikb_integrate)
    if (S->f32)
      hipLaunchKernelGGL(k_advance_q<float>, grid1(S->B), dim3(256), 0, S->stream, S->d_q, (const double*)dq, (int)shared,
                         S->nq, S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles, 0.0);
    else
      hipLaunchKernelGGL(k_advance_q<double>, grid1(S->B), dim3(256), 0, S->stream, S->d_q, (const double*)dq, (int)shared,
                         S->nq, S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles, 0.0);
    HIPCHK(hipGetLastError());
    if (!dev && !S->defer_sync) HIPCHK(hipStreamSynchronize(S->stream));  // the staging buffer is re-used by the next upload
    S->have_q = true;
  } else if (!S->have_q) {
    g_last_error = "no configurations resident on the device yet (call SolveInit / Solve with a q first)";
    return LOIKB_ERR_STATE;
  }
  if (S->f32)
    hipLaunchKernelGGL(k_fk_init<float>, grid1(S->B), dim3(256), 0, S->stream, (const double*)S->d_q, S->nq, 0,
                       S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles);
  else
    hipLaunchKernelGGL(k_fk_init<double>, grid1(S->B), dim3(256), 0, S->stream, (const double*)S->d_q, S->nq, 0,
                       S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles);
  HIPCHK(hipGetLastError());
  // the H/UDinv/Dinv cache depends on liMi; cold start: yis = 0, Aty = 0 (hxx:270-278)
  return reset_home(S, RS_HCACHE | (S->opt.warm_start ? 0 : RS_Y), S->offer_queue);
}

int constraint_products(loikb_solver_impl* S, int c_lo, int c_hi, bool grow_only)
{
  if (int frc = flush_uni(S)) return frc;
  if (S->f32)
    hipLaunchKernelGGL(k_constraint_products<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L,
                       (const float*)S->d_uni, (int)S->a_shared, c_lo, c_hi, S->B, (int)grow_only);
  else
    hipLaunchKernelGGL(k_constraint_products<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L,
                       (const double*)S->d_uni, (int)S->a_shared, c_lo, c_hi, S->B, (int)grow_only);
  HIPCHK(hipGetLastError());
  return LOIKB_OK;
}

// shared A: A and AtA computed once on the host (ik-id-description-optimized.hpp:162)
int upload_shared_A(loikb_solver_impl* S, const double* A, int c)
{
  double AtA[21];
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      double a = 0.0;
      for (int k = 0; k < 6; ++k) a += A[6 * k + i] * A[6 * k + j];
      AtA[sym(i, j)] = a;
    }
  int rc;
  if ((rc = upload_uni(S, c * 36, A, 36))) return rc;
  return upload_uni(S, S->nc * 36 + c * 21, AtA, 21);
}

std::vector<int> rowmap_constraint(const loikb_solver_impl* S, int c, int first_pair, int n)
{
  std::vector<int> rm(n);
  for (int k = 0; k < n; ++k) rm[k] = (S->L.off_c + c * S->L.crec + first_pair + k / 2) * 2 + (k & 1);
  return rm;
}

void plan_engines(loikb_solver_impl* S);

void destroy_chunks(loikb_solver_impl* S)
{
  for (Chunk& C : S->chunks) {
    if (C.h_counters) (void)hipHostFree(C.h_counters);
    if (C.d_hslots) (void)hipFree(C.d_hslots);
    if (C.d_fslots) (void)hipFree(C.d_fslots);
    if (C.d_park) (void)hipFree(C.d_park);
    if (C.d_fmask) (void)hipFree(C.d_fmask);
    if (C.d_aux) (void)hipFree(C.d_aux);
    if (C.ev_k0) (void)hipEventDestroy(C.ev_k0);
    if (C.ev_k2) (void)hipEventDestroy(C.ev_k2);
    if (C.ev_k3) (void)hipEventDestroy(C.ev_k3);
    if (C.ev_k1) (void)hipEventDestroy(C.ev_k1);
    if (C.own_stream && C.stream) (void)hipStreamDestroy(C.stream);
    void* ptrs[6] = {C.d_counters, C.d_slots, C.d_slots2, C.d_ring, C.d_order, C.d_order_bins};
    for (void* p : ptrs)
      if (p) {
        (void)hipFree(p);
        for (auto& a : S->allocs) if (a == p) a = nullptr;
      }
  }
  S->chunks.clear();
}

// chunks: contiguous ranges of tiles, each with its own stream, counters, instance lists and work queue
int build_chunks(loikb_solver_impl* S, int nchunks)
{
  const int ntiles = (S->B + WAVE - 1) / WAVE;
  const int per = (ntiles + nchunks - 1) / nchunks;
  S->chunks.clear();
  for (int t0 = 0; t0 < ntiles; t0 += per) {
    Chunk C;
    C.first_tile = t0;
    C.B = std::min(S->B - t0 * WAVE, per * WAVE);
    S->chunks.push_back(C);
  }
  void* tmp = nullptr;
  int rc;
  for (Chunk& C : S->chunks) {
    HIPCHK(hipEventCreate(&C.ev_k2));
    HIPCHK(hipEventCreate(&C.ev_k3));
    HIPCHK(hipEventCreate(&C.ev_k0));
    HIPCHK(hipEventCreate(&C.ev_k1));
    HIPCHK(hipHostMalloc((void**)&C.h_counters, NCOUNTERS * sizeof(unsigned int)));
    if ((rc = alloc_dev(S, &tmp, NCOUNTERS * sizeof(unsigned int)))) return rc;
    C.d_counters = (unsigned int*)tmp;
    if ((rc = alloc_dev(S, &tmp, sizeof(int) * (size_t)(C.B + WAVE)))) return rc;
    C.d_slots = (int*)tmp;
    if ((rc = alloc_dev(S, &tmp, sizeof(int) * (size_t)(C.B + WAVE)))) return rc;
    C.d_slots2 = (int*)tmp;
    if ((rc = alloc_dev(S, &tmp, sizeof(int) * (size_t)(C.B + WAVE)))) return rc;
    C.d_order = (int*)tmp;
    if ((rc = alloc_dev(S, &tmp, sizeof(unsigned int) * 2 * ORDER_BINS))) return rc;
    C.d_order_bins = (unsigned int*)tmp;
    C.ring_cap = 64;
    while (C.ring_cap < 2 * (C.B + WAVE)) C.ring_cap <<= 1;
    if ((rc = alloc_dev(S, &tmp, sizeof(int) * (size_t)C.ring_cap))) return rc;
    C.d_ring = (int*)tmp;
    if (S->chunks.size() > 1) {
      HIPCHK(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking));
      C.own_stream = true;
    }
  }
  return LOIKB_OK;
}

// Decade slots of the lean engine (H_i, Dinv_i per joint and decade of mu, loik_lean.hpp): sized for the whole chunk when
// the plan is made -- never inside a solve -- with a stated budget: batch x decades x 176 B x lanes per instance
// (Talos-32: 3.7 GB for 65536 instances, 59 GB for 2^20).  Not enough memory is an error the caller can act on
// (LOIKB_LEAN=0 selects the engines that need none), not a silent change of engine.
// Does the current problem allow the flat engine?  (plan.flat is the structural part: tree, precision, options; the reference
// cost is known after SolveInit / UpdateReferences: this first version of the engine serves H_ref = h I broadcast to all links)
bool href_is_scalar(const loikb_solver_impl* S)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      if (S->Href[6 * i + j] != (i == j ? S->Href[0] : 0.0)) return false;
  return true;
}
bool href_is_diagonal(const loikb_solver_impl* S)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      if (i != j && S->Href[6 * i + j] != 0.0) return false;
  return true;
}
// the builds with one instance per wavefront (k_flat2: 17..32 joints, <= FLAT_NA_SMALL ancestors; k_flat1: 33..64) also take a
// diagonal or a general reference weight shared by the links (their HM = 1 / 2 instantiations); k_flat (other trees,
// LOIKB_FLAT_SPLIT=0) takes h I only
bool flat_takes_diagonal(const loikb_solver_impl* S)
{
  if (!S->tune.flat_split || S->f32 || !S->flat.ok) return false;   // (round 4: logging handles too -- the LOG builds)
  return (S->flat.G == F2G && S->flat.nanc <= FLAT_NA_SMALL) || S->flat.G == WAVE;
}
// OSQP's rule takes mu off the decade grid: no table of slots; k_flat2 builds the factors in-wave at every change of mu
// (k_flat2<.., MUR = 1> / k_flat1<.., MUR = 1>: robots of 17..64 joints, fp64, no SolverInfo lists)
bool flat_any_mu(const loikb_solver_impl* S)
{
  return S->opt.mu_update_strat == LOIKB_MU_OSQP;
}
bool flat_any_mu_ok(const loikb_solver_impl* S)
{
  return S->tune.flat_split && !S->f32 && S->flat.ok && ((S->flat.G == F2G && S->flat.nanc <= FLAT_NA_SMALL) || S->flat.G == WAVE) && !S->opt.logging;
}
bool flat_applicable(const loikb_solver_impl* S)
{
  if (flat_any_mu(S) && !flat_any_mu_ok(S)) return false;
  // (k_flat2 / k_flat1 take any reference cost: shared or per link; k_flat h I only)
  return S->plan.flat && (flat_takes_diagonal(S) || (!S->per_link && href_is_scalar(S)));
}

// The time slice of a flat launch over n instances (0: run to completion).  End of round 4, once the SLICED builds' iteration cost
// what the plain builds' costs (it re-fetched the stopping test's tolerances from the kernel arguments every iteration): headline
// batch in arrival order 10.1 -> 9.4 ms at 288 (160 / 224 / 352 / 448: 9.6 / 9.5 / 9.5 / 9.8), 131 072: 17.5 -> 16.7, 262 144:
// 33.7 -> 32.2, 32 768: 6.1 -> 6.0, whole body 20.6 -> 19.6; 16 384: 4.16 -> 4.37 and 8192: 3.08 -> 3.32 (one straggler chain
// whatever the order: no slices below 32 768).  Not for ordered launches (their long runners start first and must not go to the
// back of the queue) and not for a handle on a stream of its own (the other batch in flight fills this one's ragged end).
constexpr int FLAT_SLICE_LATER = 96;   // (an instance's later slices: 9.44 -> 9.38 ms; a SHORT first slice -- 64 / 96 / 128, then 288 -- is slower: 9.9 / 9.6 / 9.8)
constexpr int FLAT_SLICE_DEFAULT = 288, FLAT_SLICE_MIN_BATCH = 32768, FLAT_SLICE_MAX_BATCH = 262144;   // (above: 19 KB of park record per instance for 1.5 %)
int flat_slice_for(const loikb_solver_impl* S, int n, bool ordered)
{
  if (S->tune.flat_slice >= 0) return S->tune.flat_slice | (S->tune.flat_slice2 << 16);   // (LOIKB_FLAT_SLICE[2]: as asked, whatever the launch)
  if (ordered || n < FLAT_SLICE_MIN_BATCH || n > FLAT_SLICE_MAX_BATCH || (S->opt.flags & LOIKB_OPT_OWN_STREAM) || (S->opt.flags & LOIKB_OPT_FIXED_ITERS) || S->opt.logging) return 0;
  return FLAT_SLICE_DEFAULT | (FLAT_SLICE_LATER << 16);
}

int ensure_hslots(loikb_solver_impl* S)
{
  if (S->plan.flat && S->have_problem && !flat_applicable(S)) {
    // (the problem in force cannot use the flat engine: its decade slots -- ~1 GB per 65 536 Talos instances -- go back)
    for (Chunk& C : S->chunks)
      if (C.d_fslots) { HIPCHK(hipFree(C.d_fslots)); C.d_fslots = nullptr; C.fslots_bytes = 0; }
  }
  if (S->plan.flat && (flat_applicable(S) || !S->have_problem)) {
    // decade slots of the flat engine: (ancestors + 1) scalars per lane, decade and instance
    for (Chunk& C : S->chunks) {
      const int frows = S->flat.fblk;  // (scalars per decade slot)
      const size_t need = (size_t)C.B * (flat_any_mu(S) ? 1 : S->plan.ndec) * frows * S->esz;   // (OSQP: mu0's slot only)
      if (need <= C.fslots_bytes) continue;
      if (C.d_fslots) HIPCHK(hipFree(C.d_fslots));
      C.d_fslots = nullptr; C.fslots_bytes = 0;
      if (hipMalloc(&C.d_fslots, need) != hipSuccess && !(release_getscr(S) > 0 && hipMalloc(&C.d_fslots, need) == hipSuccess)) {
        (void)hipGetLastError();
        C.d_fslots = nullptr;
        char buf[400];
        snprintf(buf, sizeof(buf), "the flat engine needs %.2f GB of decade slots for %d instances (%d decades x %d rows x %d lanes x %d B) "
                 "and the device has no room for them: create the solver with a smaller batch, or set LOIKB_FLAT=0 LOIKB_LEAN=0 to use "
                 "the k_solve + k_tail engines, which need none", need / 1e9, C.B, S->plan.ndec, frows / S->flat.G, S->flat.G, (int)S->esz);
        g_last_error = buf;
        return LOIKB_ERR_HIP;
      }
      C.fslots_bytes = need;
    }
    if (S->flat.G == F2G && flat_takes_diagonal(S) && S->tune.flat_build) {   // (the lazily populated table's per-instance masks)
      for (Chunk& C : S->chunks) {
        if ((size_t)C.B <= C.fmask_n) continue;
        if (C.d_fmask) HIPCHK(hipFree(C.d_fmask));
        C.d_fmask = nullptr; C.fmask_n = 0;
        HIPCHK(hipMalloc((void**)&C.d_fmask, sizeof(unsigned int) * (size_t)C.B));
        C.fmask_n = (size_t)C.B;
      }
    }
    // park records of the time-sliced launch (k_flat2 only: 17..32 joints): ~19 KB per instance, allocated for the batch sizes the
    // slices are used on (flat_slice_window); a handle that cannot have them simply runs unsliced
    if (S->flat.G == F2G && flat_takes_diagonal(S) && flat_slice_for(S, S->B, false) > 0) {
      for (Chunk& C : S->chunks) {
        const size_t need = (size_t)C.B * flat2_park_stride(S->nc, true) * sizeof(double);
        if (need <= C.park_bytes) continue;
        if (C.d_park) HIPCHK(hipFree(C.d_park));
        C.d_park = nullptr; C.park_bytes = 0;
        if (hipMalloc(&C.d_park, need) != hipSuccess && !(release_getscr(S) > 0 && hipMalloc(&C.d_park, need) == hipSuccess)) { (void)hipGetLastError(); C.d_park = nullptr; continue; }
        C.park_bytes = need;
      }
    }
    if (flat_applicable(S) || !S->plan.lean) return LOIKB_OK;
    if (!S->have_problem) return LOIKB_OK;  // (the lean engine's slots are only needed once a solve cannot use the flat engine)
  }
  if (!S->plan.lean) return LOIKB_OK;
  int G = 8;
  while (G < S->nb) G <<= 1;
  for (Chunk& C : S->chunks) {
    const size_t need = (size_t)C.B * S->plan.ndec * HSLOT_PAIRS * G * 2 * S->esz;
    if (need <= C.hslots_bytes) continue;
    if (C.d_hslots) HIPCHK(hipFree(C.d_hslots));
    C.d_hslots = nullptr; C.hslots_bytes = 0;
    if (hipMalloc(&C.d_hslots, need) != hipSuccess) {
      (void)hipGetLastError();
      char buf[400];
      snprintf(buf, sizeof(buf), "the lean engine needs %.2f GB of decade slots for %d instances (%d decades x %d B x %d lanes each) "
               "and the device has no room for them: create the solver with a smaller batch, or set LOIKB_LEAN=0 to use the "
               "k_solve + k_tail engines, which need none", need / 1e9, C.B, S->plan.ndec, (int)(HSLOT_PAIRS * 2 * S->esz), G);
      g_last_error = buf;
      C.d_hslots = nullptr;
      return LOIKB_ERR_HIP;
    }
    C.hslots_bytes = need;
  }
  return LOIKB_OK;
}

// The tile layout depends on whether A is shared (short constraint record), and so does the engine plan (LDS budget of
// the lean kernel, hence the number of concurrent chunks): both are (re)established here -- at create with the default
// sharing mode, at SolveInit with the real one.
int ensure_layout(loikb_solver_impl* S, bool a_shared)
{
  S->a_shared = a_shared;
  plan_engines(S);
  Layout L{};
  L.nb = S->nb; L.nc = S->nc;
  L.crec = a_shared ? CREC_SHARED_A : CREC_FULL;
  L.off_c = S->nb * JREC;
  L.off_s = L.off_c + S->nc * L.crec;
  L.tile_pairs = L.off_s + SREC;
  // odd number of 1-KiB pairs per tile: consecutive tiles (= wavefronts that run in lockstep through the same joint
  // offsets) then start on different HBM channel groups instead of camping on a few of them
  if (S->tune.tile_pad >= 0) L.tile_pairs += S->tune.tile_pad;
  else if (!(L.tile_pairs & 1)) L.tile_pairs += 1;
  const bool relayout = !S->home.tiles || L.crec != S->L.crec;
  const bool rechunk = (int)S->chunks.size() != S->plan.nchunks;
  if (!relayout && !rechunk) return ensure_hslots(S);
  auto free_set = [&](loikb_solver_impl::Set& W, bool owns_tiles) {
    void* ptrs[4] = {owns_tiles ? W.tiles : nullptr, W.map, W.wave_live, W.wave_off};
    for (void* p : ptrs)
      if (p) {
        (void)hipFree(p);
        for (auto& a : S->allocs) if (a == p) a = nullptr;
      }
    W = loikb_solver_impl::Set{};
  };
  int rc;
  if (S->home.tiles) HIPCHK(hipStreamSynchronize(S->stream));
  for (Chunk& C : S->chunks)
    for (int k = 0; k < 3; ++k) free_set(C.set[k], k != 0);
  if (rechunk) {
    destroy_chunks(S);
    if ((rc = build_chunks(S, S->plan.nchunks))) return rc;
  }
  if (relayout) {
    if (S->home.tiles) {
      // sharing mode of A changed: every tile has a different size now
      free_set(S->home, true);
      S->have_problem = false;
    }
    S->L = L;
    if ((rc = alloc_set(S, S->home, (S->B + WAVE - 1) / WAVE))) return rc;
  }
  // chunk views of the home set (their work sets are allocated on first use)
  for (Chunk& C : S->chunks) {
    loikb_solver_impl::Set& V = C.set[0];
    V.tiles = S->home.tiles + (size_t)C.first_tile * tile_bytes(S);
    V.ntiles = (C.B + WAVE - 1) / WAVE;
    void* tmp = nullptr;
    if ((rc = alloc_dev(S, &tmp, sizeof(int) * ((size_t)V.ntiles + 1)))) return rc;
    V.wave_live = (int*)tmp;
    if ((rc = alloc_dev(S, &tmp, sizeof(int) * ((size_t)V.ntiles + 1)))) return rc;
    V.wave_off = (int*)tmp;
  }
  if ((rc = ensure_hslots(S))) return rc;
  // fresh tiles: the reference ctor state is all-zero data (loik-loid-data-optimized.hxx:40-86) + ResetSolver
  return relayout ? reset_home(S, RS_SOLVER | RS_HCACHE) : LOIKB_OK;
}


// upload S->href_tab (see the member) to the device in the solve precision and in double
int upload_href_tab(loikb_solver_impl* S)
{
  const size_t n = (size_t)S->nj * HREF_ROW;
  if (!S->d_href64) {
    int rc;
    if ((rc = alloc_dev(S, (void**)&S->d_href64, n * sizeof(double)))) return rc;
    if (S->f32) { if ((rc = alloc_dev(S, &S->d_href, n * sizeof(float)))) return rc; }
    else S->d_href = S->d_href64;
  }
  // (the host vectors are members: they outlive the asynchronous copies; a second update waits for the stream first)
  HIPCHK(hipMemcpyAsync(S->d_href64, S->href_tab.data(), n * sizeof(double), hipMemcpyHostToDevice, S->stream));
  if (S->f32) {
    S->href_tab32.assign(S->href_tab.begin(), S->href_tab.end());
    HIPCHK(hipMemcpyAsync(S->d_href, S->href_tab32.data(), n * sizeof(float), hipMemcpyHostToDevice, S->stream));
  }
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

// rows of the table from one (H, v) pair per body of the caller's model ([ext_nj][36], [ext_nj][6]; stride 0 = broadcast)
void fill_href_tab(loikb_solver_impl* S, const double* H, const double* v, size_t strideH, size_t stridev)
{
  S->href_tab.assign((size_t)S->nj * HREF_ROW, 0.0);
  for (int e = 1; e < S->ext_nj; ++e) {
    double* row = S->href_tab.data() + (size_t)S->link_of[e] * HREF_ROW;  // the chain link that carries the body
    const double *He = H + strideH * e, *ve = v + stridev * e;
    memcpy(row, He, 36 * sizeof(double));
    for (int i = 0; i < 6; ++i) {
      double a = 0.0;
      for (int k = 0; k < 6; ++k) a += He[6 * i + k] * ve[k];
      row[36 + i] = a;
    }
  }
}

static bool symmetric6(const double* H)
{
  for (int i = 0; i < 6; ++i)
    for (int j = i + 1; j < 6; ++j)
      if (std::fabs(H[6 * i + j] - H[6 * j + i]) > 1e-14 * (1.0 + std::fabs(H[6 * i + j]))) return false;
  return true;
}


// joints <-> constraint slots: slot c < nc_active sits on the body active_ids[c]; the null slots behind them are parked on
// the first bodies without a constraint (the kernels find a slot through its joint; where a null slot sits changes nothing)
int bind_constraint_slots(loikb_solver_impl* S)
{
  for (int i = 1; i < S->nj; ++i) S->jd[i].cslot = -1;
  for (int c = 0; c < S->nc_active; ++c) S->jd[S->link_of[S->active_ids[c]]].cslot = c;  // the device joint that carries the body
  int e = 1;
  for (int c = S->nc_active; c < S->nc; ++c) {
    while (e < S->ext_nj && S->jd[S->link_of[e]].cslot >= 0) ++e;
    if (e >= S->ext_nj) { g_last_error = "more constraint slots than bodies"; return LOIKB_ERR_EQ_C_SIZE; }
    S->jd[S->link_of[e]].cslot = c;
  }
  S->pass_active = false;  // (the pass-level path copies the slot table when it starts)
  std::vector<int> cs(S->nj);
  for (int i = 0; i < S->nj; ++i) cs[i] = S->jd[i].cslot;
  if (cs == S->cslots_on_device) return LOIKB_OK;   // (the same task links as last time: the descriptors on the device are these)
  int rc = upload_jd(S);
  if (rc == LOIKB_OK) S->cslots_on_device = cs;
  return rc;
}

int edit_constraints(loikb_solver_impl* S, int c_lo, int c_hi, int shift)
{
  if (S->f32) hipLaunchKernelGGL(k_edit_constraints<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, c_lo, c_hi, shift, S->B);
  else hipLaunchKernelGGL(k_edit_constraints<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, c_lo, c_hi, shift, S->B);
  HIPCHK(hipGetLastError());
  return LOIKB_OK;
}

// slots [c_lo, c_hi) become null constraints
int null_constraint_slots(loikb_solver_impl* S, int c_lo, int c_hi)
{
  int rc;
  if ((rc = edit_constraints(S, c_lo, c_hi, 0))) return rc;
  const double zero[36] = {0};
  for (int c = c_lo; c < c_hi; ++c) {
    memset(S->A_host.data() + 36 * c, 0, 36 * sizeof(double));
    if ((rc = upload_shared_A(S, zero, c))) return rc;   // (harmless when A is per instance: the uniform copy is unused)
  }
  return LOIKB_OK;
}

// problem_.UpdateEqConstraint(c_id, Ai, bi), ik-id-description-optimized.hpp:178-218; Ai == NULL: the (c_id, bi) overload,
// :224-238, which keeps the old Ai
int update_eq_single(loikb_solver_impl* S, int c_id, const double* Ai, const double* bi, int in_flags)
{
  ++S->inputs_epoch;
  int found = -1, count = 0;
  for (int c = 0; c < S->nc_active; ++c)
    if (S->active_ids[c] == c_id) { if (found < 0) found = c; ++count; }
  if (found < 0) { g_last_error = loikb_status_string(LOIKB_ERR_NO_SUCH_CONSTRAINT); return LOIKB_ERR_NO_SUCH_CONSTRAINT; }
  if (count > 1) { g_last_error = loikb_status_string(LOIKB_ERR_DUP_CONSTRAINT); return LOIKB_ERR_DUP_CONSTRAINT; }
  const bool dev = in_flags & LOIKB_IN_DEVICE;
  int rc = LOIKB_OK;
  if (Ai) {
    const bool a_shared_in = in_flags & LOIKB_A_SHARED;
    if (a_shared_in != S->a_shared) {
      g_last_error = "UpdateEqConstraint: A sharing mode must match SolveInit";
      return LOIKB_ERR_ARG;
    }
    if (a_shared_in) {
      memcpy(S->A_host.data() + 36 * found, Ai, 36 * sizeof(double));
      rc = upload_shared_A(S, Ai, found);
    } else {
      rc = upload_rows(S, Ai, rowmap_constraint(S, found, CP_A, 36), dev, false);
    }
    if (rc) return rc;
  }
  if ((rc = upload_rows(S, bi, rowmap_constraint(S, found, CP_B, 6), dev, in_flags & LOIKB_B_SHARED))) return rc;
  return constraint_products(S, found, found + 1, true);
}

// problem_.UpdateReference / UpdateIneqConstraints / UpdateEqConstraints (ik-id-description-optimized.hpp:78-171,
// :325-339) with the batch layouts of loik_amd.h
// the throw sites of UpdateReference / UpdateIneqConstraints / UpdateEqConstraints, before anything of the handle changes (a rejected
// SolveInit leaves the previous problem in force, reference weight and plan included: ADVICE r04)
int validate_problem(const loikb_solver_impl* S, const double* H_ref, const int* c_ids, int nc, int nbound)
{
  if (nbound != S->nv) { g_last_error = "lb/ub dimension differs from model.nv"; return LOIKB_ERR_INEQ_DIM; }
  if (nc != S->nc_active) { g_last_error = "number of equality constraints doesn't match initialization"; return LOIKB_ERR_EQ_C_SIZE; }
  if (!symmetric6(H_ref)) { g_last_error = "H_ref must be symmetric"; return LOIKB_ERR_HREF_NOT_SYMMETRIC; }
  for (int c = 0; c < nc; ++c) {
    if (c_ids[c] < 1 || c_ids[c] >= S->ext_nj) { g_last_error = "constraint link id out of range"; return LOIKB_ERR_ARG; }
    for (int c2 = 0; c2 < c; ++c2)
      if (c_ids[c2] == c_ids[c]) { g_last_error = "multiple constraints on the same link"; return LOIKB_ERR_DUP_CONSTRAINT; }
  }
  return LOIKB_OK;
}

int set_problem(loikb_solver_impl* S, const double* H_ref, const double* v_ref, const int* c_ids, int nc,
                const double* Ais, const double* bis, const double* lb, const double* ub, int nbound, int in_flags)
{
  if (int vrc = validate_problem(S, H_ref, c_ids, nc, nbound)) return vrc;
  ++S->inputs_epoch;
  const bool dev = in_flags & LOIKB_IN_DEVICE;
  int rc;
  // UpdateReference: Hv = H_ref v_ref, Hv_inf_norm_ (hpp:85-96)
  // (what the TABLE holds, not what S->Href holds: SolveInit stores the new weight there before the plan is made)
  const bool same_ref = S->d_href64 && S->tab_bcast && !memcmp(S->tab_H, H_ref, sizeof(S->tab_H)) && !memcmp(S->tab_v, v_ref, sizeof(S->tab_v));
  memcpy(S->Href, H_ref, sizeof(S->Href));
  memcpy(S->vref, v_ref, sizeof(S->vref));
  S->href_diag = true;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      if (i != j && H_ref[6 * i + j] != 0.0) S->href_diag = false;
  S->Hv_inf_norm = 0.0;
  for (int i = 0; i < 6; ++i) {
    double a = 0.0;
    for (int k = 0; k < 6; ++k) a += H_ref[6 * i + k] * v_ref[k];
    S->Hv[i] = a;
    if (std::fabs(a) > S->Hv_inf_norm) S->Hv_inf_norm = std::fabs(a);
  }
  S->per_link = false;
  if (!same_ref) {  // (the table the getters and the pass-level path read; the engines take the broadcast pair as arguments)
    fill_href_tab(S, H_ref, v_ref, 0, 0);
    S->tab_bcast = false;
    if ((rc = upload_href_tab(S))) return rc;
    memcpy(S->tab_H, H_ref, sizeof(S->tab_H));
    memcpy(S->tab_v, v_ref, sizeof(S->tab_v));
    S->tab_bcast = true;
  }
  // UpdateIneqConstraints
  S->bnd_shared = in_flags & LOIKB_BOUNDS_SHARED;
  if (S->bnd_shared) {
    if ((rc = upload_uni(S, S->nc * 57, lb, S->nv))) return rc;
    if ((rc = upload_uni(S, S->nc * 57 + S->nb, ub, S->nv))) return rc;
  } else {
    std::vector<int> rl(S->nb), ru(S->nb);
    for (int j = 0; j < S->nb; ++j) { rl[j] = (j * JREC + JP_LBUB) * 2; ru[j] = rl[j] + 1; }
    if ((rc = upload_rows(S, lb, rl, dev, false))) return rc;
    if ((rc = upload_rows(S, ub, ru, dev, false))) return rc;
  }
  // UpdateEqConstraints
  S->active_ids.assign(c_ids, c_ids + nc);
  if ((rc = bind_constraint_slots(S))) return rc;
  S->a_shared = in_flags & LOIKB_A_SHARED;
  S->A_host.assign((size_t)S->nc * 36, 0.0);
  if (nc < S->nc && (rc = null_constraint_slots(S, nc, S->nc))) return rc;
  for (int c = 0; c < nc; ++c) {
    if (S->a_shared) {
      memcpy(S->A_host.data() + 36 * c, Ais + 36 * c, 36 * sizeof(double));
      if ((rc = upload_shared_A(S, Ais + 36 * c, c))) return rc;
    }
  }
  if (!S->a_shared && nc > 0) {
    // (instance-major [B][nc][36]: the rows of the active slots; with spare slots the tile records are not contiguous per
    //  instance in the caller's array, which the row map takes care of)
    std::vector<int> rm;
    for (int c = 0; c < nc; ++c) { auto r = rowmap_constraint(S, c, CP_A, 36); rm.insert(rm.end(), r.begin(), r.end()); }
    if ((rc = upload_rows(S, Ais, rm, dev, false))) return rc;
  }
  if (nc > 0) {
    std::vector<int> rm;
    for (int c = 0; c < nc; ++c) { auto r = rowmap_constraint(S, c, CP_B, 6); rm.insert(rm.end(), r.begin(), r.end()); }
    if ((rc = upload_rows(S, bis, rm, dev, in_flags & LOIKB_B_SHARED))) return rc;
  }
  if ((rc = constraint_products(S, 0, S->nc, false))) return rc;
  S->have_problem = true;
  return ensure_hslots(S);  // (which engine's decade slots this problem needs is known now: H_ref)
}

// instances of the home set that ran out of iterations: finished, neither converged nor flagged infeasible
template <typename T>
__global__ void k_count_unfinished(char* tiles, Layout L, int B, unsigned int* __restrict__ counter)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  int hit = 0;
  if (b < B) {
    char* sp = lane_ptr<T>(tiles, L, b);
    const int status = (int)ldp<T>(sp + (size_t)L.off_s * pair_bytes<T>(), SP_ST).x;
    hit = !(status & (ST_CONVERGED | ST_PRIMAL_INF));
  }
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & (WAVE - 1)) == 0 && m) atomicAdd(counter, (unsigned int)__popcll(m));
}

// move the live instances of set `src` (n_src slots) to the first slots of set `dst`; finished ones go home.
// dst < 0: end of the solve, everything still in a work set goes home.
template <typename T>
int compact(loikb_solver_impl* S, Chunk* C, int src, int dst, int n_src, int* n_dst_out)
{
  loikb_solver_impl::Set& A = C->set[src];
  const int nw = (n_src + WAVE - 1) / WAVE;
  int rc;
  if (dst >= 0 && (rc = alloc_set(S, C->set[dst], C->set[0].ntiles, C->stream, true))) return rc;
  // exclusive scan of the per-wavefront live counts on the host (nw <= B/64 ints)
  C->h_wave.resize(2 * (size_t)nw + 2);
  int* cnt = C->h_wave.data();
  int* off = cnt + nw + 1;
  int total = 0;
  if (dst >= 0) {
    HIPCHK(hipMemcpyAsync(cnt, A.wave_live, sizeof(int) * nw, hipMemcpyDeviceToHost, C->stream));
    HIPCHK(hipStreamSynchronize(C->stream));
    for (int w = 0; w < nw; ++w) { off[w] = total; total += cnt[w]; }
    HIPCHK(hipMemcpyAsync(A.wave_off, off, sizeof(int) * nw, hipMemcpyHostToDevice, C->stream));
  }
  MovePlan M{};
  M.src = A.tiles;
  M.dst_live = dst >= 0 ? C->set[dst].tiles : nullptr;
  M.dst_home = src == 0 ? nullptr : C->set[0].tiles;
  M.L = S->L;
  M.n_src = n_src;
  M.move_bounds = !S->bnd_shared;
  M.map_src = src == 0 ? nullptr : A.map;
  M.map_dst = dst >= 0 ? C->set[dst].map : nullptr;
  M.wave_off = A.wave_off;
  M.force_home = dst < 0;
  hipLaunchKernelGGL(k_move<T>, dim3(nw), dim3(WAVE), 0, C->stream, M);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(C->stream));  // h_wave is reused
  if (n_dst_out) *n_dst_out = total;
  return LOIKB_OK;
}

// finish the remaining live instances of set `cur` (n_cur slots, n_live of them live) with the cooperative tail
// kernel (a lane group per instance, one joint per lane)
// Can the stragglers / small batches of this solver run in the lean tail kernel (two wavefronts per SIMD, loik_lean.hpp)?
// H cache on, one joint per lane, at most 4 children per joint, and an LDS footprint that
// lets two 4-wavefront workgroups share a CU.  LOIKB_LEAN=0 switches it off.
// wavefronts of the lean kernel a CU can hold: two per SIMD by registers (256 each), fewer when the LDS of 160 KB is the
// limit (each wavefront owns its exchange rows, one H slot per lane and its instances' constraint blocks: 19.5 KB with one
// shared-A task constraint, 21.6 KB with four).  Wavefronts of this kernel never synchronise with each other, so the
// workgroup size is free: four wavefronts per workgroup while eight fit, single-wavefront workgroups otherwise (seven fit
// with four task constraints -- a whole-body task set stays in this engine at 7/8 of the residency).
int lean_waves_per_cu(const loikb_solver_impl* S)
{
  int G = 8;
  while (G < S->nb) G <<= 1;
  const size_t per_wave = S->f32 ? lean_lds_bytes<float>(S->nc, G, S->a_shared) : lean_lds_bytes<double>(S->nc, G, S->a_shared);
  return (int)std::min<size_t>(8, (160 * 1024) / per_wave);
}

// May this launch take the set in the order the handle's previous solve left (k_order_*)?  Yes when the iteration counts were taken
// on the inputs this solve has (exact: a cold Solve() of the same problem repeats itself); when the inputs changed since, only for a
// caller that declared consecutive problems alike (LOIKB_OPT_ORDER_FROM_PREVIOUS), and then subject to the hold-off the timing
// comparison sets -- a stale order that predicts nothing is arrival order with the long runners in random places, a few per cent
// slower than arrival order itself.
static bool order_usable(const loikb_solver_impl* S, const Chunk* C, int n_cur)
{
  if (!S->tune.flat_order || C->order_n != n_cur) return false;   // (round 4: handles on a stream of their own are ordered too)
  if (C->order_epoch == S->inputs_epoch) return true;
  return (S->opt.flags & LOIKB_OPT_ORDER_FROM_PREVIOUS) && C->order_holdoff == 0;
}

// THE engine dispatch: (nb, nc, A shared?, children per joint, precision, options, tuning) -> plan.  Called from
// ensure_layout: at create with the default sharing mode of A, at SolveInit with the real one.
void plan_engines(loikb_solver_impl* S)
{
  EnginePlan pl;
  pl.ndec = S->tune.lean_decades;
  pl.kexp_lo = S->tune.lean_klo;
  if (S->opt.flags & LOIKB_OPT_FIXED_ITERS) { pl.ndec = 1; pl.kexp_lo = 0; }  // mu frozen at mu0: one decade
  pl.lean_waves_cu = S->nb <= WAVE ? lean_waves_per_cu(S) : 0;
  // Single-wavefront workgroups: wavefronts of k_lean never synchronise with each other, and a workgroup gives its registers and
  // LDS back as soon as ITS two instances are done -- with four wavefronts per workgroup one long runner kept the slots of
  // seven finished neighbours, which costs nothing while one launch owns the machine (headline 21.85 vs 21.99 ms, noise) but
  // is what a second batch in flight has to wait for (two batches in flight: 32.3 vs 35.6 ms per pair).
  pl.lean_wg_waves = 1;
  if (S->tune.lean_wg_waves > 0 && pl.lean_waves_cu % S->tune.lean_wg_waves == 0) pl.lean_wg_waves = S->tune.lean_wg_waves;
  // (configurations that never reach the on-chip engines must not plan -- and allocate decade slots for -- them)
  const char* never = S->opt.logging ? "logging = 1: every solve runs on the pass-by-pass implementation"
                      : S->opt.tail_max_instances < 0 ? "tail_max_instances < 0: the caller asked for the solve kernel alone"
                      : (S->opt.flags & LOIKB_OPT_NO_COMPACTION) ? "LOIKB_OPT_NO_COMPACTION: the solve kernel keeps every instance in its tile" : nullptr;
  if (never) pl.why_not_lean = never;
  else if (!S->tune.lean) pl.why_not_lean = "LOIKB_LEAN=0";
  else if (S->nb > WAVE) pl.why_not_lean = "more joints than lanes of a wavefront";
  else if (S->maxchild > 4) pl.why_not_lean = "a joint with more than four children";
  else if (S->opt.flags & LOIKB_OPT_NO_H_CACHE) pl.why_not_lean = "LOIKB_OPT_NO_H_CACHE (no precomputed H)";
  else if (S->opt.mu_update_strat == LOIKB_MU_OSQP) pl.why_not_lean = "OSQP penalty rule: mu is off the decade grid";
  // Eight wavefronts per CU is what the registers allow; seven (single-wavefront workgroups) when four constraint blocks -- a
  // whole-body task set -- take more LDS.  Robots of up to 16 joints pack 4-8 instances into a wavefront and solve in a handful
  // of iterations: for them k_solve + k_tail are faster in every case measured (Panda-7, B = 65536, fp64 / fp32: tol 1e-3
  // 0.75 / 0.61 ms against 1.18 / 1.14 ms in k_lean, tol 1e-4 0.89 / 0.81 against - / 1.33; the ten precomputed decades of H
  // are mostly never used by such short solves) -- scripts/r02/small_robot_plan_probe.py.
  // (... up to the batch k_tail takes whole, 32 768 instances.  Above it the choice is k_solve + a hand-over to k_tail against k_lean
  //  for the whole batch, and since k_lean takes a handle's later solves longest first it wins: Panda-7, B = 65 536, tol 1e-3 / 1e-4,
  //  fp64 0.73 / 0.92 -> 0.60 / 0.66 ms, fp32 0.68 / 0.80 -> 0.52 / 0.55 ms; a handle's first solve 1.3 -> 1.5 ms)
  else if (S->nb <= 16 && !(S->f32 && (S->opt.flags & LOIKB_OPT_F32_ACCURATE)) && S->B <= 32768)
    pl.why_not_lean = "a small robot (<= 16 joints): k_solve + k_tail are faster on its short solves";
  else if (pl.lean_waves_cu < 7) pl.why_not_lean = "constraint blocks leave too few wavefronts per CU in LDS";
  else pl.lean = true;
  // the flat engine (no loops over the tree levels, loik_flat.hpp): same regime as k_lean, any number of children per joint
  if (S->flat.ok) {
    const size_t per_wave = S->flat.nanc <= FLAT_NA_SMALL ? flat_lds_bytes<double, FLAT_NA_SMALL>(S->nc, S->flat.G, S->a_shared, false)
                                                           : flat_lds_bytes<double, FLAT_MAXA>(S->nc, S->flat.G, S->a_shared, false);
    pl.flat_waves_cu = (int)std::min<size_t>(8, (160 * 1024) / per_wave);
    // (the builds with one instance per wavefront have their own LDS layouts: what counts is the kernel that would run)
    if (flat_takes_diagonal(S)) {
      const size_t pw = S->flat.G == F2G ? flat2_lds_bytes<FLAT_NA_SMALL>(S->nc, true)
                        : S->flat.nanc <= FLAT_NA_SMALL ? flat1_lds_bytes<FLAT_NA_SMALL>(S->nc, true, S->tune.flat_one_slot ? 1 : 2)
                                                        : flat1_lds_bytes<FLAT_MAXA>(S->nc, true, S->tune.flat_one_slot ? 1 : 2);
      pl.flat_waves_cu = std::max(pl.flat_waves_cu, (int)std::min<size_t>(8, (160 * 1024) / pw));
    }
  }
  // (logging = 1 does not keep a solve off the flat engines: their LOG builds write the SolverInfo lists themselves)
  const char* never_flat = S->opt.tail_max_instances < 0 ? never : (S->opt.flags & LOIKB_OPT_NO_COMPACTION) ? never : nullptr;
  if (never_flat) pl.why_not_flat = never_flat;
  else if (!S->tune.flat) pl.why_not_flat = S->tune.lean ? "LOIKB_FLAT=0" : "LOIKB_LEAN=0";
  else if (!S->flat.ok) pl.why_not_flat = S->flat.why;
  else if (S->f32) pl.why_not_flat = "fp32 solver";
  else if (S->opt.flags & LOIKB_OPT_NO_H_CACHE) pl.why_not_flat = "LOIKB_OPT_NO_H_CACHE (no precomputed factors)";
  else if (flat_any_mu(S) && !flat_any_mu_ok(S))
    pl.why_not_flat = "OSQP penalty rule: mu is off the decade grid, and the in-wave builder is k_flat2's / k_flat1's (17..64 joints, fp64, no logging)";
  else if (S->nb <= 16) pl.why_not_flat = "a small robot (<= 16 joints): k_solve + k_tail are faster on its short solves";
  // (the flat engines update the task constraints on lanes 6 c + k of an instance's lanes, in one pass: ten constraints with a
  //  wavefront per instance, five with two instances per wavefront; more go to the engines that loop over them)
  else if (S->nc > (flat_takes_diagonal(S) ? 10 : S->flat.G / 6)) pl.why_not_flat = "more task constraints than the flat engine's lanes update in one pass";
  else if (pl.flat_waves_cu < (flat_takes_diagonal(S) ? 4 : 6)) pl.why_not_flat = "constraint blocks leave too few wavefronts per CU in LDS";
  else pl.flat = true;
  // with the lean kernel whole batches up to 2^20 instances go to it directly (it is as fast as k_solve's bulk phase and
  // has neither ragged tiles nor compaction); without it k_solve hands over to k_tail at 32768 live instances
  // (pl.flat is structural; whether THIS problem can use the flat engine also depends on its reference weight -- k_flat, the build
  //  other trees and LOIKB_FLAT_SPLIT=0 run, takes H_ref = h I only.  A handle that has neither engine for its problem hands over to
  //  k_tail at 32 768 live instances and solves in two chunks, as before the on-chip engines: ADVICE r03)
  const bool flat_usable = pl.flat && (!S->href_known || flat_takes_diagonal(S) || href_is_scalar(S));
  pl.tail_max = S->opt.tail_max_instances > 0 ? S->opt.tail_max_instances : ((pl.lean || flat_usable) ? (1 << 20) : 32768);
  // Concurrent chunks pay only in the k_solve + k_tail configuration (measured on MI355X, Talos-32, B = 65536: 1 chunk
  // 52.9 ms/step, 2 chunks 47.9, 3 chunks 48.2, 4 chunks 74: one chunk's latency-bound straggler phase runs beside the
  // other's bulk phase); the lean kernel takes the whole batch in one launch.
  const int ntiles = (S->B + WAVE - 1) / WAVE;
  pl.nchunks = (ntiles >= 512 && !pl.lean && !flat_usable) ? 2 : 1;
  if (S->tune.chunks > 0) pl.nchunks = S->tune.chunks;
  pl.nchunks = std::max(1, std::min(pl.nchunks, ntiles));
  // (k_solve's LDS: edge slots of the leaf->root sweeps -- aliased by the team's scalar exchange -- + v slots; as in run_chunk)
  {
    auto need_of = [&](const loikb_solver_impl::TeamSched& sc) {
      const int edge = std::max(std::max(sc.nslots, 1) * EDGE_ENT, sc.nw > 1 ? sc.nw * Norms<double>::NALL : 0);
      return (size_t)(edge + std::max(sc.nvslots, 1) * 6) * WAVE * S->esz;
    };
    pl.solve_lds_need = need_of(S->sched[0]);   // (the team schedule is optional: run_chunk uses it only when it fits)
    pl.solve_ok = pl.solve_lds_need <= (size_t)160 * 1024;
    if (!pl.solve_ok) { pl.tail_max = 1 << 20; pl.nchunks = 1; }   // (whole batches to the on-chip engines)
  }
  S->plan = pl;
}

// a robot k_solve cannot take (EnginePlan::solve_ok): does this solve go to the on-chip engines whole?  (run_chunk's direct_tail)
static bool bushy_goes_on_chip(const loikb_solver_impl* S)
{
  return S->nb <= WAVE && S->opt.tail_max_instances >= 0 && S->opt.max_launch_iters <= 0 &&
         !(S->opt.flags & (LOIKB_OPT_NO_COMPACTION | LOIKB_OPT_NO_H_CACHE)) && S->B <= S->plan.tail_max && S->tune.direct_tail;
}

template <typename T>
int run_tail(loikb_solver_impl* S, Chunk* C, Params<T>& P, int cur, int n_cur, int n_live, double* ms_out,
             unsigned long long* iters_out, bool whole_set = false)
{
  loikb_solver_impl::Set& A = C->set[cur];
  const int nw = (n_cur + WAVE - 1) / WAVE;
  // Small batches on the one-instance-per-wavefront flat engines (every instance gets a wavefront at once: no schedule to prepare, no order
  // to leave for the next solve): the list, the ring and the counters come from ONE kernel, the order pass is skipped, and n_unfinished is
  // counted by k_list_unfinished -- 8 launches / copies and one synchronisation less per Solve() (B = 1: 0.168 -> see profiles/r06_*_small_batches)
  int G0 = 8;
  while (G0 < S->nb) G0 <<= 1;
  const bool small_flat = whole_set && cur == 0 && S->chunks.size() == 1 && sizeof(T) == 8 && S->tune.flat_split && flat_applicable(S) && (P.mode & MODE_CACHE_H) &&
                          ((G0 == F2G && S->flat.ok && S->flat.nanc <= FLAT_NA_SMALL) || G0 == WAVE) && n_cur <= S->tune.flat_small_batch &&
                          n_cur >= std::max(1, S->tune.flat_min_batch);
  if (whole_set && small_flat) {
    // (k_queue_init_iota below, once the ring's size is known to the flat path)
  } else if (whole_set) {
    // every slot of the set (finished instances, if any, stop at once inside the kernel)
    hipLaunchKernelGGL(k_list_iota, grid1(n_cur), dim3(256), 0, C->stream, C->d_slots, n_cur);
  } else {
    C->h_wave.resize(2 * (size_t)nw + 2);
    int* cnt = C->h_wave.data();
    int* off = cnt + nw + 1;
    HIPCHK(hipMemcpyAsync(cnt, A.wave_live, sizeof(int) * nw, hipMemcpyDeviceToHost, C->stream));
    HIPCHK(hipStreamSynchronize(C->stream));
    int total = 0;
    for (int w = 0; w < nw; ++w) { off[w] = total; total += cnt[w]; }
    if (total != n_live) { g_last_error = "tail: live count mismatch"; return LOIKB_ERR_STATE; }
    HIPCHK(hipMemcpyAsync(A.wave_off, off, sizeof(int) * nw, hipMemcpyHostToDevice, C->stream));
    hipLaunchKernelGGL(k_list_live<T>, dim3(nw), dim3(WAVE), 0, C->stream, A.tiles, S->L, n_cur, A.wave_off, C->d_slots);
  }
  HIPCHK(hipGetLastError());
  Bufs<T> Bf = make_bufs<T>(S, C, cur);
  P.B = n_cur;
  int G = 8;  // lanes per instance: smallest power of two >= nb
  while (G < S->nb) G <<= 1;
  const int ipw = WAVE / G;
  int tw = S->tune.tail_waves;  // wavefronts per workgroup
  while (tw > 1 && tw * tail_lds_bytes<T>(S->nc, G) > 160 * 1024) --tw;
  const size_t lds = tw * tail_lds_bytes<T>(S->nc, G);
  if (lds > 160 * 1024) { g_last_error = "tail kernel: constraint data exceeds the LDS of a CU"; return LOIKB_ERR_ARG; }
  if (lds > 64 * 1024) {
    HIPCHK(hipFuncSetAttribute((const void*)k_tail<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute((const void*)k_tail<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // ONE launch: as many workgroups as the chunk's share of the CUs holds (one wavefront per SIMD: register budget);
  // the lane groups pull the listed instances from an atomic queue head until the list is empty.
  const int cu_share = std::max(1, (int)(S->ncu * (TAIL_WAVES / tw) * ((double)C->B / (double)S->B) + 0.5));
  int n = n_live;
  double total_ms = 0.0;
  unsigned long long iters = 0;
  const bool trace = S->tune.trace;
  const int* list = C->d_slots;
  // ---- lean tail kernel: two wavefronts per SIMD (loik_lean.hpp).  H_i / Dinv_i / UDinv_i of the listed instances are
  // precomputed for the decades mu0 * 10^(0 .. ndec-1) (k_hslots); instances whose mu leaves them come back unfinished
  // and go through k_tail below.
  {
    // decades of mu with precomputed slots: mu0 * 10^(kexp_lo .. kexp_lo + ndec - 1).  The DEFAULT rule moves mu up from
    // mu0 in the first iterations and then mostly oscillates between two or three decades (Talos workload: 0..7 seen,
    // < 0 never); an instance that leaves the range is finished by k_tail.
    // (the table is built as a pipeline over the tree levels: a decade more costs one step)
    // Every decade costs k_hslots a pipeline step and 176 B x joints per instance of HBM writes (0.14 ms per decade on the
    // headline): after the first solve of a handle the table covers the decades its instances were seen in, plus one on
    // each side (the headline workload lives in 0..6 of the configured -2..7: 21.8 -> 21.2 ms).  mu restarts at mu0 in every
    // solve, batches of one application resemble each other; when they do not, the instance that leaves the table escapes
    // to k_tail as always and the full range is back for the next solve.
    int ndec = S->plan.ndec, kexp_lo = S->plan.kexp_lo;
    // (a logged solve keeps the whole configured range: an instance that escaped to k_tail would stop writing its lists)
    if (S->tune.lean_adapt && S->seen_hi >= S->seen_lo && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS) && !S->opt.logging) {
      const int lo = std::max(kexp_lo, S->seen_lo - 1), hi = std::min(kexp_lo + ndec - 1, S->seen_hi + 1);
      if (hi >= lo) { kexp_lo = lo; ndec = hi - lo + 1; }
    }
    const int ndec_hist = ndec, kexp_lo_hist = kexp_lo;
    const size_t wave_lds = lean_lds_bytes<T>(S->nc, G, S->a_shared);
    // ---- the flat engine (loik_flat.hpp: no loops over the tree levels) takes the place of k_hslots + k_lean when the solve's
    // reference cost allows (H_ref = h I for all links)
    // (the builds with ONE instance per wavefront -- k_flat2, k_flat1 -- take any number of instances: a lone instance iterates in 2.3 us
    //  there against k_tail's 10-12 us, and the reference's own call is one problem, tests/loik-loid.cpp:987-1032; k_flat, two instances
    //  per wavefront, from 64 as ever)
    const bool flat_one_per_wave = S->tune.flat_split && sizeof(T) == 8 && ((G == F2G && S->flat.ok && S->flat.nanc <= FLAT_NA_SMALL) || G == WAVE);
    const bool flat_ok = flat_applicable(S) && (P.mode & MODE_CACHE_H) && n >= (flat_one_per_wave ? std::max(1, S->tune.flat_min_batch) : 64);
    if (flat_ok) {
      const int nanc = S->flat.nanc;
      const bool small_na = nanc <= FLAT_NA_SMALL;
      const int frows = S->flat.fblk;  // (scalars per decade slot of an instance, packed columns: loik_flat.hpp)
      // the rule that moves mu (k_flat2's MUR): 1 = OSQP's -- no table, every change of mu is an in-wave build --, 2 = decade steps with
      // the in-wave builder for the decades the table lacks (LOIKB_FLAT_BUILD=1), 0 = decade steps, the table or k_tail
      // (k_flat2 only: for k_flat1 -- 33..64 joints -- it was built and measured in round 6 and did not pay: loik_flat2.hpp)
      const bool can_build2 = S->tune.flat_build && G == F2G && small_na && sizeof(T) == 8 && S->tune.flat_split && C->d_fmask != nullptr &&
                              !S->opt.logging && !S->per_link && href_is_scalar(S);
      int mur = flat_any_mu(S) ? 1 : (can_build2 ? 2 : 0);
      if (mur == 1) { ndec = 1; kexp_lo = 0; }   // (the table holds mu0's slot: where every cold solve starts)
      if (mur == 2) { ndec = S->plan.ndec; kexp_lo = S->plan.kexp_lo; }   // (the lazily populated table keeps its whole range: addresses, not work)
      const size_t need = (size_t)n_cur * ndec * frows * sizeof(T);
      if (need > C->fslots_bytes) { g_last_error = "internal: decade-slot buffer of the flat engine smaller than the chunk"; return LOIKB_ERR_STATE; }
      const int has_hv = S->Hv_inf_norm != 0.0;
      const size_t flds = small_na ? flat_lds_bytes<T, FLAT_NA_SMALL>(S->nc, G, S->a_shared, has_hv) : flat_lds_bytes<T, FLAT_MAXA>(S->nc, G, S->a_shared, has_hv);
      int lgG = 3;
      while ((1 << lgG) < G) ++lgG;
      const double cu_sh = std::max(1.0, S->ncu * ((double)C->B / (double)S->B));
      const int cap_lat = (int)std::min<size_t>(4, (160 * 1024) / flds) * (int)(cu_sh + 0.5);  // k_flat: one wavefront per SIMD
      // two lanes per joint where the layout applies: 17..32 joints, few ancestors, fp64, no lists to write
      const bool split = S->tune.flat_split && G == F2G && small_na && sizeof(T) == 8;
      // one instance per wavefront anyway (33..64 joints): the build with nested loops, prefix-sum subtree sums, DPP fold
      const bool one = S->tune.flat_split && G == WAVE && sizeof(T) == 8;
      P.max_launch_iters = S->opt.max_iter + 1;
      const int mode_keep = P.mode;
      if (whole_set && S->zero_state && S->tune.flat_zero_state) P.mode |= MODE_ZERO_STATE;
      // (queue_ready: this solve's reset launch left list, ring and counters -- nothing lies between the solve's first event and this point)
      const bool queue_ready = small_flat && C->queue_ready && n_cur == S->B;
      C->queue_ready = false;
      if (queue_ready) {}
      else if (small_flat) hipLaunchKernelGGL(k_queue_init_iota, grid1(std::max(C->ring_cap, NCOUNTERS)), dim3(256), 0, C->stream, C->d_ring, C->ring_cap, C->d_slots, n_cur, C->d_counters, NCOUNTERS);
      else HIPCHK(hipMemsetAsync(C->d_counters, 0, NCOUNTERS * sizeof(unsigned int), C->stream));
      const hipEvent_t ev_first = queue_ready ? S->ev_t0 : C->ev_k0;
      if (!queue_ready) HIPCHK(hipEventRecord(C->ev_k0, C->stream));
      // longest first: the order the previous solve of this handle left (k_order_*, loik_lean.hpp) when this launch takes the same
      // whole set; the decade slots are indexed by the INSTANCE slot (sidx / lidx), so k_fslots and the engine find them under any order of the list
      const bool ordered = !small_flat && whole_set && list == C->d_slots && n == n_cur && (split || one) && order_usable(S, C, n_cur);
      const bool order_was_stale = ordered && C->order_epoch != S->inputs_epoch;
      if (C->order_holdoff > 0) --C->order_holdoff;
      if (ordered) HIPCHK(hipMemcpyAsync(C->d_slots, C->d_order, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, C->stream));
      C->stats.flat_ordered += ordered ? 1 : 0;
      // The decades k_fslots builds now: all of the table, or (MUR = 2) a window of it -- the rest is populated by the instances that get
      // there (k_flat2<.., MUR = 2>, loik_flat2.hpp).  The window: LOIKB_FLAT_WINDOW=lo,n as given (any launch); else, for a time-sliced
      // launch of a handle with a history, from the decade its previous solve's instances STARTED in (0 after a cold reset) to the one
      // 97 % of them had ENDED in or below (k_order_count's histogram) -- the headline's instances end in decades 0..3 (8.7 / 23.8 / 60.6 /
      // 6.9 %) of the 0..6 the handle's history says were visited by somebody: four decades built instead of nine, k_fslots 0.9 -> 0.55 ms,
      // and the 0.2 % that go further build their slot once each (70 us).  Without a window the launch runs the build without the builder
      // (MUR = 0: its iteration loop is 1-3 % shorter).
      int dw0 = 0, nw = ndec;
      if (mur == 2) {
        int wlo = 0, whi = -1;
        // (whether this launch is time-sliced, decided ONCE: the park buffer's size included -- ADVICE r05)
        int q = (split && !S->opt.logging) ? flat_slice_for(S, n, ordered) : 0;
        if (q > 0 && (C->d_park == nullptr || (size_t)n_cur * flat2_park_stride(S->nc, true) * sizeof(double) > C->park_bytes)) q = 0;
        const bool win_ok = q > 0 && C->d_park != nullptr;   // (a window on time-sliced launches only: the builder beside the unsliced loop costs more than k_fslots saves)
        if (S->tune.flat_win_n > 0) { wlo = S->tune.flat_win_lo; whi = wlo + S->tune.flat_win_n - 1; }
        else if (S->tune.flat_build == 2 && win_ok && S->end_hist_n > 0) {
          unsigned long long cum = 0;
          int k03 = 99, k97 = 99;
          for (int k = 0; k < 32; ++k) {
            cum += S->end_hist[k];
            if (k03 == 99 && cum * 100 >= 3ull * S->end_hist_n) k03 = k - 16;
            if (k97 == 99 && cum * 100 >= 97ull * S->end_hist_n) k97 = k - 16;
          }
          wlo = std::min(0, k03); whi = std::max(0, k97);
        }
        else if (S->tune.flat_build == 2 && win_ok && n >= 49152) {
          // No history (a handle's first solve): the five decades from mu0's upwards.  The rule moves mu up far more often than down, and
          // rarely more than three decades; whoever leaves the window builds the slot once (results are bit-identical whatever the window:
          // tests/test_engines.py::test_flat_lazy_table_*).  Measured, time-sliced headline launches, ms per solve incl. k_fslots, full
          // table / this window: 65 536 instances 8.67 / 8.32, 131 072: 14.96 / 14.64, 262 144: 29.16 / 28.63; 32 768 and below no gain
          // (5.45 / 5.49), nor on unsliced launches (the builder beside an unsliced loop costs it more than k_fslots saves:
          // 65 536 ordered 7.44 / 7.59): profiles/r05_i_lazy_window_sizes.txt
          wlo = 0; whi = 4;
        }
        const int lo = std::max(wlo, kexp_lo), hi = std::min(whi, kexp_lo + ndec - 1);
        if (hi >= lo && (hi - lo + 1) < ndec) { dw0 = lo - kexp_lo; nw = hi - lo + 1; }
        else if (S->tune.flat_build == 2) mur = 0;   // (auto: nothing to leave out -- the build without the builder)
      }
      if (mur == 0) {   // (the range the handle's history asks for, as ever)
        ndec = ndec_hist; kexp_lo = kexp_lo_hist;
        nw = ndec; dw0 = 0;
      }
      const int win_bits = (int)(((nw >= 16 ? 0xFFFFu : ((1u << nw) - 1u)) << dw0) & 0xFFFFu);
      if (ndec > 0) {
        const dim3 hgrid((unsigned)((n + ipw - 1) / ipw));
        // [65][22] exchange rows (pass B's L columns, [NA + 1][64], live in them afterwards) + [65][6] S^w + the constraints' A^T A + the decades' mu
        const size_t slds = (std::max((size_t)(WAVE + 1) * 22, (size_t)((small_na ? FLAT_NA_SMALL : FLAT_MAXA) + 1) * WAVE) + (size_t)(WAVE + 1) * 6 +
                             (size_t)ipw * S->nc * 21 + 16) * sizeof(T);
        if (small_na)
          hipLaunchKernelGGL((k_fslots<T, FLAT_NA_SMALL>), hgrid, dim3(WAVE), slds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                             (const TailTopo*)S->d_topo, (const int*)S->d_child_list, (const FlatLane*)S->flat.d_lanes, S->maxdepth,
                             nanc, frows, S->flat.njmp, list, n, G, (T*)C->d_fslots, kexp_lo, ndec, S->tune.fslot_dgrp > 0 ? S->tune.fslot_dgrp : nw,
                             dw0, nw, mur == 2 ? C->d_fmask : (unsigned int*)nullptr);
        else
          hipLaunchKernelGGL((k_fslots<T, FLAT_MAXA>), hgrid, dim3(WAVE), slds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                             (const TailTopo*)S->d_topo, (const int*)S->d_child_list, (const FlatLane*)S->flat.d_lanes, S->maxdepth,
                             nanc, frows, S->flat.njmp, list, n, G, (T*)C->d_fslots, kexp_lo, ndec, S->tune.fslot_dgrp > 0 ? S->tune.fslot_dgrp : nw,
                             dw0, nw, mur == 2 ? C->d_fmask : (unsigned int*)nullptr);
        HIPCHK(hipGetLastError());
      }
      const bool slot_event = !small_flat || S->tune.small_slot_event || S->tune.flat_probe > 0;   // (the probe launch's time is taken from this event)
      if (slot_event) HIPCHK(hipEventRecord(C->ev_k2, C->stream));
      const int n_first = n;
      bool probe_timed = false, small_finish = false;
      dim3 grid((unsigned)std::min((n + ipw - 1) / ipw, cap_lat));
      {
        if (!small_flat) hipLaunchKernelGGL(k_ring_fill, grid1(C->ring_cap), dim3(256), 0, C->stream, C->d_ring, C->ring_cap, list, n, C->d_counters);
        if (split || one) {   // (what only the in-wave builder / the lazily populated table read, behind one kernel argument)
          // (uploaded when one of the pointers changes -- synchronously, from the chunk's own copy: ADVICE r05)
          if (C->d_aux == nullptr) HIPCHK(hipMalloc((void**)&C->d_aux, 4 * sizeof(void*)));
          const void* auxh[4] = {S->d_topo, S->d_child_list, C->d_fmask, nullptr};
          if (memcmp(auxh, C->aux_h, sizeof(auxh)) != 0) {
            memcpy(C->aux_h, auxh, sizeof(auxh));
            HIPCHK(hipStreamSynchronize(C->stream));
            HIPCHK(hipMemcpy(C->d_aux, C->aux_h, sizeof(auxh), hipMemcpyHostToDevice));
          }
        }
        if (split) {
          // k_flat2: two lanes per joint, one instance per wavefront, two or three wavefronts per SIMD (loik_flat2.hpp)
          const size_t lds2 = flat2_lds_bytes<FLAT_NA_SMALL>(S->nc, has_hv != 0, mur == 1);
          const int wpe = S->tune.flat_split_wpe;
          int per_cu = (int)std::min<size_t>((size_t)4 * wpe, (160 * 1024) / lds2);
          if (S->tune.lean_wg_per_cu > 0) per_cu = std::min(per_cu, S->tune.lean_wg_per_cu);
          grid = dim3((unsigned)std::min(n, per_cu * (int)(cu_sh + 0.5)));
          // Round-robin time slicing inside the launch (k_flat2<.., SLICED>).  Iteration counts are heavy-tailed and unknown: run to
          // completion in arrival order, the 999-iteration instances that are fetched late keep the launch alive ~3 ms after
          // the queue ran dry.  With a slice of 160 iterations every long runner has done ~500 by then.  Measured (Talos-32,
          // one GPU, ms per batch without / with): 8192: 4.23 / 4.39, 16 384: 5.46 / 5.61, 32 768: 8.13 / 7.80, 65 536: 13.28 /
          // 12.29, 131 072: 23.13 / 22.24, 262 144: 44.19 / 44.48 -- a switch costs a store, an agent-scope reload and the set-up of
          // an instance (~25 us), small batches are one straggler chain whatever the order, large ones hide it in their bulk:
          // on by default between 12 and 96 instances per resident wavefront.  Slices of 32 / 64 / 96 / 128 / 192 / 256 / 384
          // iterations on the headline: 19.6 / 13.7 / 12.40 / 12.36 / 12.33 / 12.45 / 12.76 ms.
          const int resident = (int)grid.x;
          // (an ordered launch runs to completion: its long runners start first and must not go to the back of the queue.  Time
          //  slices for everything behind the predicted-long prefix were tried: the SLICED build's agent-scope loads / stores of
          //  the records cost the short instances more than the slices bring -- 12.2 ms against 10.6)
          // Round 4: OFF by default.  A sliced-out instance is now parked (its lane state and LDS blocks in a lane-contiguous record of
          // its own, ~40 row accesses each way, the ticket for the next entry drawn with the stores, the decade slot fetched with the
          // record: three round trips instead of ten through the tile records) -- and with the iteration at 2.9 instead of 3.6 us the
          // slices still bring only 0..2.5 %: headline batch in arrival order, k_flat2 alone, ms: unsliced 9.47, slices of 192 / 128 /
          // 96 / 64 / 48 / 32: 9.32 / 9.28 / 9.32 / 9.76 / 11.45 / 13.9; 131 072 instances 16.73 -> 16.31 (128); 32 768: 5.87 -> 5.82;
          // 16 384 and below: slower (one straggler chain whatever the order).  A switch costs ~27 us of a wavefront under load (five
          // dependent trips to the L2 / HBM at ~2.5 us each when 2048 wavefronts share them), a slice of 64 iterations 190 us.
          // LOIKB_FLAT_SLICE=q switches it on.
          // End of round 4: ON again for arrival-order launches of >= 32 768 instances (flat_slice_for has the numbers).
          const size_t park_need = (size_t)n_cur * flat2_park_stride(S->nc, true) * sizeof(double);
          (void)resident;
          int quantum = flat_slice_for(S, n, ordered);
          if (quantum > 0 && (C->d_park == nullptr || park_need > C->park_bytes)) quantum = 0;
          // (not for a handle on a stream of its own: that is how batches are kept in flight side by side, and then the other
          //  batch's bulk fills this one's ragged end -- slicing only adds its switches, and its wavefronts that wait for queue
          //  entries hold slots the other launch could use: two headline batches in flight 21.5 ms per pair without, 24.9 with)
#define LOIKB_LAUNCH_FLAT2(WPE, ...)                                                                                            \
  hipLaunchKernelGGL((k_flat2<FLAT_NA_SMALL, WPE, ##__VA_ARGS__>), grid, dim3(WAVE), lds2, C->stream,                            \
                     *reinterpret_cast<const Params<double>*>(&P), *reinterpret_cast<const Bufs<double>*>(&Bf),                  \
                     (const JointDesc*)S->d_jd, (const FlatLane*)S->flat.d_lanes, nanc, S->flat.nscan, S->flat.njmp, C->d_ring, n, \
                     (const double*)C->d_fslots, frows, kexp_lo, ndec, (double)S->Href[0], (int)((unsigned int)has_hv | ((unsigned int)(S->maxdepth & 0xFF) << 8) | ((unsigned int)win_bits << 16)), C->ring_cap - 1, quantum,       \
                     (double*)C->d_park, flat2_park_stride(S->nc, true), (const void* const*)C->d_aux)
          const int hm = S->per_link ? 3 : href_is_scalar(S) ? 0 : href_is_diagonal(S) ? 1 : 2;
          // Two launches instead of one time-sliced one (FLAT_Q_PROBE / FLAT_Q_FINISH, loik_flat2.hpp): the probe, k_probe_sort, the survivors.
          const int probe_len = S->tune.flat_probe, probe_mark = std::min(S->tune.flat_probe_mark, std::max(1, probe_len / 2));
          const bool two_launches = quantum > 0 && probe_len >= 2 && !ordered && mur != 1 && !S->opt.logging && wpe != 3 &&
                                    S->opt.max_iter > probe_len && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS);
          const int quantum_one = quantum;
          auto launch_flat2 = [&](int quantum) {
          if (mur == 1) {
            // (OSQP's rule runs unsliced unless LOIKB_FLAT_SLICE asks: a parked instance rebuilds its factors when it is taken up again --
            //  headline batch, first solve: 12.7 ms with the default slices, 11.4 without: profiles/r05_d_mu_rules.jsonl)
            if (S->tune.flat_slice < 0) quantum = 0;
            if (hm == 3) { quantum = 0; LOIKB_LAUNCH_FLAT2(2, false, 3, false, 1); }
            else if (hm == 2) { quantum = 0; LOIKB_LAUNCH_FLAT2(2, false, 2, false, 1); }
            else if (hm == 1) { quantum = 0; LOIKB_LAUNCH_FLAT2(2, false, 1, false, 1); }
            else if (quantum > 0) LOIKB_LAUNCH_FLAT2(2, true, 0, false, 1);
            else LOIKB_LAUNCH_FLAT2(2, false, 0, false, 1);
          }
          else if (mur == 2) {
            if (quantum > 0) LOIKB_LAUNCH_FLAT2(2, true, 0, false, 2); else LOIKB_LAUNCH_FLAT2(2, false, 0, false, 2);
          }
          else if (S->opt.logging) {   // (the SolverInfo lists: unsliced; a diagonal reference weight goes as a general one)
            quantum = 0;
            if (hm == 3) LOIKB_LAUNCH_FLAT2(2, false, 3, true);
            else if (hm >= 1) LOIKB_LAUNCH_FLAT2(2, false, 2, true);
            else LOIKB_LAUNCH_FLAT2(2, false, 0, true);
          }
          else if (hm == 3) { if (quantum > 0) LOIKB_LAUNCH_FLAT2(2, true, 3); else LOIKB_LAUNCH_FLAT2(2, false, 3); }
          else if (hm == 2) { if (quantum > 0) LOIKB_LAUNCH_FLAT2(2, true, 2); else LOIKB_LAUNCH_FLAT2(2, false, 2); }
          else if (hm == 1) { if (quantum > 0) LOIKB_LAUNCH_FLAT2(2, true, 1); else LOIKB_LAUNCH_FLAT2(2, false, 1); }
          else if (quantum > 0) { if (wpe == 3) LOIKB_LAUNCH_FLAT2(3, true); else LOIKB_LAUNCH_FLAT2(2, true); }
          else if (wpe == 3) LOIKB_LAUNCH_FLAT2(3);
          else LOIKB_LAUNCH_FLAT2(2);
          };
          if (two_launches) {
            launch_flat2(probe_mark | ((probe_len - probe_mark) << 16) | FLAT_Q_PROBE);
            HIPCHK(hipGetLastError());
            const int nl_ = (has_hv ? F2G * 6 : 0) + S->nc * C2D;   // (the record's LDS blocks: shv | constraint blocks | the scalar block)
            hipLaunchKernelGGL(k_probe_sort, dim3(1), dim3(1024), 0, C->stream, C->d_ring, n, C->ring_cap - 1, (const double*)C->d_park,
                               flat2_park_stride(S->nc, true), FLAT2_PARK_ROWS * WAVE + nl_, FLAT2_PARK_ROWS * WAVE + nl_ + FISC, C->d_counters,
                               probe_mark, probe_len - probe_mark > FLAT_PROBE_LAST ? probe_len - FLAT_PROBE_LAST : probe_len, probe_len, S->opt.max_iter,
                               (double)S->opt.tol_abs);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(C->ev_k3, C->stream));
            launch_flat2(FLAT_Q_FINISH);
            C->stats.flat_probe_launches++;
            probe_timed = true;
          } else {
            launch_flat2(quantum_one);
          }
#undef LOIKB_LAUNCH_FLAT2
        } else if (one) {
          // one decade slot in LDS instead of two when that buys wavefronts per CU (whole body, four task constraints: 6 -> 8)
          auto lds_of = [&](int bufs) { return small_na ? flat1_lds_bytes<FLAT_NA_SMALL>(S->nc, has_hv != 0, bufs) : flat1_lds_bytes<FLAT_MAXA>(S->nc, has_hv != 0, bufs); };
          // (OSQP's rule: the in-wave builder's rows lie over both slots -- two it is)
          const int one_buf = mur != 1 && (S->tune.flat_one_slot != 0) && std::min<size_t>(8, (160 * 1024) / lds_of(1)) > std::min<size_t>(8, (160 * 1024) / lds_of(2));
          const size_t lds1 = lds_of(one_buf ? 1 : 2);
          int per_cu1 = (int)std::min<size_t>(8, (160 * 1024) / lds1);
          if (S->tune.lean_wg_per_cu > 0) per_cu1 = std::min(per_cu1, S->tune.lean_wg_per_cu);
          grid = dim3((unsigned)std::min(n, per_cu1 * (int)(cu_sh + 0.5)));
          // (time slicing as in k_flat2, same window: whole body, four tasks, B = 65 536: 34.7 ms without)
          const int resident = (int)grid.x, full = per_cu1 * (int)(cu_sh + 0.5);
          (void)resident; (void)full;
          const int quantum = mur == 1 ? 0 : flat_slice_for(S, n, ordered);
#define LOIKB_LAUNCH_FLAT1(NAV, ...)                                                                                            \
  hipLaunchKernelGGL((k_flat1<NAV, ##__VA_ARGS__>), grid, dim3(WAVE), lds1, C->stream,                                          \
                     *reinterpret_cast<const Params<double>*>(&P), *reinterpret_cast<const Bufs<double>*>(&Bf),                  \
                     (const JointDesc*)S->d_jd, (const FlatLane*)S->flat.d_lanes, nanc, S->flat.nscan, S->flat.njmp, C->d_ring, n, \
                     (const double*)C->d_fslots, frows, kexp_lo, ndec, (double)S->Href[0], has_hv | (one_buf ? 2 : 0) | ((S->maxdepth & 0xFF) << 8),  \
                     C->ring_cap - 1, quantum, (const void* const*)C->d_aux)
          const int hm = S->per_link ? 3 : href_is_scalar(S) ? 0 : href_is_diagonal(S) ? 1 : 2;
          if (mur == 1) {   // (OSQP's rule: unsliced; a diagonal reference weight goes as a general one)
            if (hm == 3) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 3, false, 1); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 3, false, 1); }
            else if (hm >= 1) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 2, false, 1); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 2, false, 1); }
            else if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 0, false, 1);
            else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 0, false, 1);
          }
          else if (S->opt.logging) {   // (as k_flat2's)
            const int quantum = 0;
            if (hm == 3) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 3, true); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 3, true); }
            else if (hm >= 1) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 2, true); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 2, true); }
            else if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 0, true);
            else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 0, true);
          }
          else if (hm == 3) {
            if (quantum > 0) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, true, 3); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, true, 3); }
            else if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 3);
            else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 3);
          } else if (hm == 2) {
            if (quantum > 0) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, true, 2); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, true, 2); }
            else if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 2);
            else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 2);
          } else if (hm == 1) {
            if (quantum > 0) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, true, 1); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, true, 1); }
            else if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, false, 1);
            else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, false, 1);
          } else if (quantum > 0) { if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL, true); else LOIKB_LAUNCH_FLAT1(FLAT_MAXA, true); }
          else if (small_na) LOIKB_LAUNCH_FLAT1(FLAT_NA_SMALL);
          else LOIKB_LAUNCH_FLAT1(FLAT_MAXA);
#undef LOIKB_LAUNCH_FLAT1
        } else {
#define LOIKB_LAUNCH_FLAT(NAV, ...)                                                                                             \
  hipLaunchKernelGGL((k_flat<T, NAV, ##__VA_ARGS__>), grid, dim3(WAVE), flds, C->stream, P, Bf, (const JointDesc*)S->d_jd,                \
                     (const FlatLane*)S->flat.d_lanes, nanc, S->flat.nscan, S->flat.njmp, (const int*)C->d_ring, n, lgG,          \
                     (const T*)C->d_fslots, frows, kexp_lo, ndec, (T)S->Href[0], has_hv)
        if (S->opt.logging) { if (small_na) LOIKB_LAUNCH_FLAT(FLAT_NA_SMALL, true); else LOIKB_LAUNCH_FLAT(FLAT_MAXA, true); }
        else if (small_na) LOIKB_LAUNCH_FLAT(FLAT_NA_SMALL);
        else LOIKB_LAUNCH_FLAT(FLAT_MAXA);
#undef LOIKB_LAUNCH_FLAT
        }
        HIPCHK(hipGetLastError());
        P.mode = mode_keep;
        int* nxt = (list == C->d_slots) ? C->d_slots2 : C->d_slots;
        small_finish = small_flat && S->tune.small_finish;
        if (small_finish)   // (one workgroup: the list, the counts, and the launch's counters into the pinned host copy -- loik_lean.hpp)
          hipLaunchKernelGGL(k_small_finish<T>, dim3(1), dim3(256), 0, C->stream, A.tiles, S->L, list, n, nxt, C->d_counters, NCOUNTERS, C->h_counters);
        else
          hipLaunchKernelGGL(k_list_unfinished<T>, grid1(n), dim3(256), 0, C->stream, A.tiles, S->L, list, n, nxt, C->d_counters + 3);
        HIPCHK(hipGetLastError());
        C->stats.launches++;
        C->stats.tail_launches++;
        C->stats.lean_launches++;
        C->stats.flat_launches++;
        if (split) C->stats.flat_split_launches++;
      }
      int* next = (list == C->d_slots) ? C->d_slots2 : C->d_slots;
      HIPCHK(hipEventRecord(C->ev_k1, C->stream));
      const bool order_pass = !small_flat && S->tune.flat_order && whole_set && n_first == n_cur && (split || one) && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS);
      if (!order_pass && !small_finish) HIPCHK(hipMemcpyAsync(C->h_counters, C->d_counters, NCOUNTERS * sizeof(unsigned int), hipMemcpyDeviceToHost, C->stream));
      if (order_pass) {
        // the order for the handle's next solve: longest first by the iteration counts of this one (an instance that escaped to
        // k_tail counts with what it had when it left); and the decades the instances ended in (the next sliced launch's table window)
        HIPCHK(hipMemsetAsync(C->d_order_bins, 0, sizeof(unsigned int) * 2 * ORDER_BINS, C->stream));
        hipLaunchKernelGGL(k_order_count<T>, grid1(n_cur), dim3(256), 0, C->stream, A.tiles, S->L, n_cur, S->opt.max_iter, C->d_order_bins,
                           C->d_counters + ORDER_DEC_HIST);
        HIPCHK(hipMemcpyAsync(C->h_counters, C->d_counters, NCOUNTERS * sizeof(unsigned int), hipMemcpyDeviceToHost, C->stream));
        hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_BINS), 0, C->stream, C->d_order_bins);
        hipLaunchKernelGGL(k_order_scatter<T>, grid1(n_cur), dim3(256), 0, C->stream, A.tiles, S->L, n_cur, S->opt.max_iter, C->d_order_bins, C->d_order);
        HIPCHK(hipGetLastError());
        C->order_n = n_cur;
        C->order_epoch = S->inputs_epoch;
      }
      HIPCHK(hipStreamSynchronize(C->stream));
      float ms = 0.f, t0 = 0.f, hms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ev_first, C->ev_k1));
      if (!queue_ready) HIPCHK(hipEventElapsedTime(&t0, S->ev_t0, C->ev_k0));
      if (slot_event) HIPCHK(hipEventElapsedTime(&hms, ev_first, C->ev_k2));
      iters += C->h_counters[1];
      if (C->h_counters[FLAT_COUNTERS_ERR]) {
        g_last_error = (C->h_counters[FLAT_COUNTERS_ERR] & 4u) ? "internal: the flat engine's dynamic LDS does not start at LDS address 0 (lds_abs)"
                                                               : "internal: a wavefront of the flat engine gave up waiting on its work queue";
        return LOIKB_ERR_STATE;
      }
      C->stats.lean_requeues += (int)C->h_counters[LEAN_Q_REQUEUES];
      const unsigned int escaped = C->h_counters[2];
      C->stats.flat_built += (int)C->h_counters[FLAT_COUNTERS_BUILT];
      {
        std::lock_guard<std::mutex> lock(S->alloc_mu);
        const unsigned int seen = C->h_counters[LEAN_DECADES_SEEN];
        for (int d = 0; d < 16; ++d)
          if (seen & (1u << d)) { S->seen_lo = std::min(S->seen_lo, kexp_lo + d); S->seen_hi = std::max(S->seen_hi, kexp_lo + d); }
        if (escaped) { S->seen_lo = S->plan.kexp_lo; S->seen_hi = S->plan.kexp_lo + S->plan.ndec - 1; }
        if (order_pass) {   // (summed over the chunks of one solve: the first chunk to report starts the histogram anew -- ADVICE r05)
          if (S->end_hist_epoch != S->solve_serial) { S->end_hist_epoch = S->solve_serial; S->end_hist_n = 0; for (int k = 0; k < 32; ++k) S->end_hist[k] = 0; }
          for (int k = 0; k < 32; ++k) { S->end_hist[k] += C->h_counters[ORDER_DEC_HIST + k]; S->end_hist_n += C->h_counters[ORDER_DEC_HIST + k]; }
        }
        if (mur == 2) {   // (the decades somebody took up, loaded or built: the handle's history as the table's range sees it)
          for (int k = 0; k < 32; ++k)
            if (C->h_counters[FLAT_COUNTERS_DEC + k]) { S->seen_lo = std::min(S->seen_lo, k - 16); S->seen_hi = std::max(S->seen_hi, k - 16); }
        }
        S->ud_stale = true;
      }
      C->stats.hslots_ms += hms;
      if (probe_timed) { float pms = 0.f; HIPCHK(hipEventElapsedTime(&pms, C->ev_k2, C->ev_k3)); C->stats.probe_ms += pms; }
      if (C->h_counters[FLAT_COUNTERS_DRY])  // (100 MHz clock, low words: from the ring fill of the last stage to the first empty fetch)
        C->stats.queue_dry_ms += (double)(unsigned int)(C->h_counters[FLAT_COUNTERS_TDRY] - C->h_counters[FLAT_COUNTERS_T0]) * 1e-5;
      if ((split || one) && whole_set && n_first == n_cur) {
        // did the order predict this solve?  The launch is compared with the handle's last launch in arrival order (+ time slices)
        // of the same set: a batch that resembles the previous one runs 5..35 % shorter ordered (no ragged end); one that does
        // not is arrival order without the slices that would have softened it, a few per cent LONGER -- then the next four solves
        // go back to arrival order, which also refreshes the figure to compare with.  (What the launch leaves after its queue ran
        // dry does not tell: a small batch's queue is empty long before its long runners are done, whatever the order.)
        const double flat_ms = (double)ms - (double)hms;
        if (C->arrival_n != n_cur) { C->arrival_n = n_cur; C->arrival_ms = 0.0; }
        if (!ordered) C->arrival_ms = flat_ms;
        else if (order_was_stale && C->arrival_ms > 0.0 && flat_ms > 0.985 * C->arrival_ms) C->order_holdoff = S->tune.flat_order_holdoff;
      }
      if (trace)
        fprintf(stderr, "[loikb] flat engine: %6d instances on %u workgroups, done at %8.3f ms (slots %6.3f ms)  "
                        "inst-iters %9u  wave-iters %7u  slot loads %7u (+ %u served from LDS)  escaped %u  still iterating %u\n",
                n_first, grid.x, ms, hms, C->h_counters[1], C->h_counters[5], C->h_counters[6],
                C->h_counters[FLAT_COUNTERS_SLOT_HITS], escaped, C->h_counters[3]);
      C->stats.lean_escaped += (int)escaped;
      C->tail_iv.emplace_back(t0, t0 + ms);
      total_ms = ms;
      n = (int)C->h_counters[3];
      list = next;
      // (the launch took the chunk's whole home set and nothing is still iterating: k_list_unfinished has counted loikb_stats::n_unfinished)
      if (n == 0 && whole_set && cur == 0 && n_first == n_cur && S->chunks.size() == 1) { C->stats.n_unfinished = (int)C->h_counters[4]; C->unfinished_counted = true; C->small_finished = small_finish; }
      if (n == 0) { *ms_out = total_ms; *iters_out = iters; return LOIKB_OK; }
      // what is left escaped the precomputed decades: k_tail below finishes it
    }
    // (k_lean's lane groups need whole wavefronts of work to pay: below 64 instances k_tail's direct path is as good)
    const bool lean_ok = !flat_ok && S->plan.lean && (P.mode & MODE_CACHE_H) && n >= 64;
    const bool per_link = S->per_link;  // (UpdateReferences' table: the PERLINK instantiations of k_hslots / k_lean)
    if (lean_ok) {
      // decade slots are indexed by the instance's slot in the set (relaunches with shorter lists find them again);
      // the buffer was sized for the chunk at SolveInit (ensure_hslots)
      const size_t need = (size_t)n_cur * ndec * HSLOT_PAIRS * G * 2 * sizeof(T);
      if (need > C->hslots_bytes) { g_last_error = "internal: decade-slot buffer smaller than the chunk"; return LOIKB_ERR_STATE; }
    }
    if (lean_ok) {
      const int waves_cu = S->plan.lean_waves_cu;
      const int ltw = S->plan.lean_wg_waves;  // wavefronts per workgroup
      const size_t lds = ltw * wave_lds;
      if (lds > 64 * 1024) {
        HIPCHK(hipFuncSetAttribute((const void*)k_lean<T, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_lean<T, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_lean<T, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_lean<T, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_lean<T, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_lean<T, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      }
      const int wg_per_cu = S->tune.lean_wg_per_cu > 0 ? S->tune.lean_wg_per_cu : waves_cu / ltw;
      const int wg_cap = wg_per_cu * std::max(1, (int)(S->ncu * ((double)C->B / (double)S->B) + 0.5));
      // Optional rounds with a bounded share of iterations per instance (LOIKB_LEAN_QUANTA="24,64,160,400"; default: one
      // launch).  Iteration counts are heavy-tailed and unpredictable: in one launch the work queue drains after ~60 % of
      // the launch time and the rest is waiting for long runners that were fetched late.  Bounded rounds make every long
      // runner advance from the start -- but it then only advances `quantum` iterations per round, i.e. it shares the
      // machine instead of running flat out from an early start.  Measured: Talos headline (1.2 % of the instances run
      // all 1000 iterations) 22.85 -> 21.99 ms, B = 131072 36.9 -> 36.2 ms; floating-base Talos (a handful of long runners)
      // 36.5 -> 44.1 ms.  Not a default.
      std::vector<int> quanta = S->tune.lean_quanta;
      quanta.push_back(S->opt.max_iter + 1);
      // (default 0: run to completion in arrival order -- DESIGN.md, scheduling study)
      const int lean_quantum = S->tune.lean_slice;
      HIPCHK(hipEventRecord(C->ev_k0, C->stream));
      // longest first, as in the flat engine: the order the handle's previous solve left, for a single whole-set launch that is not
      // time-sliced (the decade slots are indexed by the instance: k_hslots and k_lean see the same list)
      const bool lean_ordered = whole_set && list == C->d_slots && n == n_cur && quanta.size() == 1 && lean_quantum == 0 && order_usable(S, C, n_cur);
      const bool lean_order_was_stale = lean_ordered && C->order_epoch != S->inputs_epoch;
      if (whole_set && C->order_holdoff > 0) --C->order_holdoff;
      if (lean_ordered) HIPCHK(hipMemcpyAsync(C->d_slots, C->d_order, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, C->stream));
      C->stats.flat_ordered += lean_ordered ? 1 : 0;
      const int n_first_lean = n;
      float t_first = -1.f;
      unsigned int escaped = 0;
      bool slots_built = false, first_timed = false;
      for (size_t round = 0; round < quanta.size() && n > 0; ++round) {
        if (quanta[round] <= 0) continue;
        P.max_launch_iters = quanta[round];
        const int wg_needed = (n + ipw * ltw - 1) / (ipw * ltw);
        const dim3 grid((unsigned)std::min(wg_needed, wg_cap));
        HIPCHK(hipMemsetAsync(C->d_counters, 0, NCOUNTERS * sizeof(unsigned int), C->stream));
        if (!slots_built) {
          slots_built = true;
          const dim3 hgrid((unsigned)((n + ipw - 1) / ipw));
          const size_t hlds = (size_t)(WAVE + 1) * 22 * sizeof(T);
          if (per_link)
            hipLaunchKernelGGL((k_hslots<T, false, true>), hgrid, dim3(WAVE), hlds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                               (const TailTopo*)S->d_topo, (const int*)S->d_child_list, S->maxdepth, S->maxchild, list, n, G,
                               (T*)C->d_hslots, kexp_lo, ndec);
          else if (S->href_diag)
            hipLaunchKernelGGL((k_hslots<T, true>), hgrid, dim3(WAVE), hlds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                               (const TailTopo*)S->d_topo, (const int*)S->d_child_list, S->maxdepth, S->maxchild, list, n, G,
                               (T*)C->d_hslots, kexp_lo, ndec);
          else
            hipLaunchKernelGGL((k_hslots<T, false>), hgrid, dim3(WAVE), hlds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                               (const TailTopo*)S->d_topo, (const int*)S->d_child_list, S->maxdepth, S->maxchild, list, n, G,
                               (T*)C->d_hslots, kexp_lo, ndec);
          HIPCHK(hipEventRecord(C->ev_k2, C->stream));
        }
        hipLaunchKernelGGL(k_ring_fill, grid1(C->ring_cap), dim3(256), 0, C->stream, C->d_ring, C->ring_cap, list, n, C->d_counters);
        // time slice of the in-kernel round-robin queue (0 = run every instance to completion in arrival order); bounded
        // host-side rounds (LOIKB_LEAN_QUANTA) bring their own bound and switch it off
        const int quantum = quanta.size() > 1 ? 0 : lean_quantum;
#define LOIKB_LAUNCH_LEAN(HD, SL, ...)                                                                                        \
  hipLaunchKernelGGL((k_lean<T, HD, SL, ##__VA_ARGS__>), grid, dim3(WAVE * ltw), lds, C->stream, P, Bf, (const JointDesc*)S->d_jd,             \
                     (const TailTopo*)S->d_topo, (const int*)S->d_child_list, S->maxdepth, S->maxchild, C->d_ring,              \
                     C->ring_cap - 1, n, G, (const T*)C->d_hslots, kexp_lo, ndec, quantum, S->multi_from)
        if (per_link) { if (quantum > 0) LOIKB_LAUNCH_LEAN(false, true, true); else LOIKB_LAUNCH_LEAN(false, false, true); }
        else if (quantum > 0) { if (S->href_diag) LOIKB_LAUNCH_LEAN(true, true); else LOIKB_LAUNCH_LEAN(false, true); }
        else { if (S->href_diag) LOIKB_LAUNCH_LEAN(true, false); else LOIKB_LAUNCH_LEAN(false, false); }
#undef LOIKB_LAUNCH_LEAN
        HIPCHK(hipGetLastError());
        // the instances that are still iterating: the next round's list (ping-pong between the two list buffers)
        int* next = (list == C->d_slots) ? C->d_slots2 : C->d_slots;
        hipLaunchKernelGGL(k_list_unfinished<T>, grid1(n), dim3(256), 0, C->stream, A.tiles, S->L, list, n, next, C->d_counters + 3);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(C->ev_k1, C->stream));
        HIPCHK(hipMemcpyAsync(C->h_counters, C->d_counters, NCOUNTERS * sizeof(unsigned int), hipMemcpyDeviceToHost, C->stream));
        const bool lean_orders = S->tune.flat_order && whole_set && n_first_lean == n_cur && quanta.size() == 1 && lean_quantum == 0 &&
                                 !(S->opt.flags & LOIKB_OPT_FIXED_ITERS);
        if (lean_orders) {  // the order for the handle's next solve (k_order_*)
          HIPCHK(hipMemsetAsync(C->d_order_bins, 0, sizeof(unsigned int) * 2 * ORDER_BINS, C->stream));
          hipLaunchKernelGGL(k_order_count<T>, grid1(n_cur), dim3(256), 0, C->stream, A.tiles, S->L, n_cur, S->opt.max_iter, C->d_order_bins);
          hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_BINS), 0, C->stream, C->d_order_bins);
          hipLaunchKernelGGL(k_order_scatter<T>, grid1(n_cur), dim3(256), 0, C->stream, A.tiles, S->L, n_cur, S->opt.max_iter, C->d_order_bins, C->d_order);
          HIPCHK(hipGetLastError());
          C->order_n = n_cur;
          C->order_epoch = S->inputs_epoch;
        }
        HIPCHK(hipStreamSynchronize(C->stream));
        float ms = 0.f, t0 = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, C->ev_k0, C->ev_k1));  // since the start of the first round
        if (lean_orders) {  // (held off like the flat engine's: compared with the handle's last launch in arrival order)
          if (C->arrival_n != n_cur) { C->arrival_n = n_cur; C->arrival_ms = 0.0; }
          if (!lean_ordered) C->arrival_ms = (double)ms;
          else if (lean_order_was_stale && C->arrival_ms > 0.0 && (double)ms > 0.985 * C->arrival_ms) C->order_holdoff = S->tune.flat_order_holdoff;
        }
        if (t_first < 0.f) { HIPCHK(hipEventElapsedTime(&t0, S->ev_t0, C->ev_k0)); t_first = t0; }
        iters += C->h_counters[1];
        escaped = C->h_counters[2];
        {
          std::lock_guard<std::mutex> lock(S->alloc_mu);
          const unsigned int seen = C->h_counters[LEAN_DECADES_SEEN];
          for (int d = 0; d < 16; ++d)
            if (seen & (1u << d)) { S->seen_lo = std::min(S->seen_lo, kexp_lo + d); S->seen_hi = std::max(S->seen_hi, kexp_lo + d); }
          if (escaped) { S->seen_lo = S->plan.kexp_lo; S->seen_hi = S->plan.kexp_lo + S->plan.ndec - 1; }  // the table was too narrow
        }
        C->stats.lean_requeues += (int)C->h_counters[LEAN_Q_REQUEUES];
        if (!first_timed) {
          first_timed = true;
          float hms = 0.f;
          HIPCHK(hipEventElapsedTime(&hms, C->ev_k0, C->ev_k2));
          C->stats.hslots_ms += hms;
        }
        if (trace)
          fprintf(stderr, "[loikb] lean tail round %zu (<= %d iterations each): %6d instances on %u workgroups, done at %8.3f ms"
                          "  inst-iters %9u  wave-iters %7u slot loads %7u  escaped %u  still iterating %u  requeued %u\n",
                  round, quanta[round], n, grid.x, ms, C->h_counters[1], C->h_counters[5], C->h_counters[6], escaped,
                  C->h_counters[3], C->h_counters[LEAN_Q_REQUEUES]);
        C->stats.launches++;
        C->stats.tail_launches++;
        C->stats.lean_launches++;
        total_ms = ms;
        n = (int)C->h_counters[3];
        list = next;
      }
      C->tail_iv.emplace_back(t_first, t_first + (float)total_ms);
      C->stats.lean_escaped += (int)escaped;
      if (n == 0) { *ms_out = total_ms; *iters_out = iters; return LOIKB_OK; }
      // what is left escaped the precomputed decades: k_tail below finishes it
    }
  }
  {
    P.max_launch_iters = S->opt.max_iter + 1;
    const int wg_needed = (n + ipw * tw - 1) / (ipw * tw);
    const dim3 grid((unsigned)std::min(wg_needed, cu_share));
    HIPCHK(hipMemsetAsync(C->d_counters, 0, NCOUNTERS * sizeof(unsigned int), C->stream));
    HIPCHK(hipEventRecord(C->ev_k0, C->stream));
    // k_tail as the engine of a whole batch (OSQP rule, robots outside the on-chip engines' domain): longest first from the handle's
    // previous solve, as in the flat and lean engines (its lane groups pull the list in order)
    const bool tail_whole = whole_set && list == C->d_slots && n == n_cur && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS) &&
                            S->tune.flat_order;
    const bool tail_ordered = tail_whole && order_usable(S, C, n_cur);
    const bool tail_order_was_stale = tail_ordered && C->order_epoch != S->inputs_epoch;
    if (tail_whole && C->order_holdoff > 0) --C->order_holdoff;
    if (tail_ordered) HIPCHK(hipMemcpyAsync(C->d_slots, C->d_order, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, C->stream));
    C->stats.flat_ordered += tail_ordered ? 1 : 0;
    if (S->href_diag)
      hipLaunchKernelGGL((k_tail<T, true>), grid, dim3(WAVE * tw), lds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                         (const TailTopo*)S->d_topo, (const int*)S->d_child_list, S->maxdepth, S->maxchild,
                         list, n, G);
    else
      hipLaunchKernelGGL((k_tail<T, false>), grid, dim3(WAVE * tw), lds, C->stream, P, Bf, (const JointDesc*)S->d_jd,
                         (const TailTopo*)S->d_topo, (const int*)S->d_child_list, S->maxdepth, S->maxchild,
                         list, n, G);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(C->ev_k1, C->stream));
    HIPCHK(hipMemcpyAsync(C->h_counters, C->d_counters, NCOUNTERS * sizeof(unsigned int), hipMemcpyDeviceToHost, C->stream));
    if (tail_whole) {
      HIPCHK(hipMemsetAsync(C->d_order_bins, 0, sizeof(unsigned int) * 2 * ORDER_BINS, C->stream));
      hipLaunchKernelGGL(k_order_count<T>, grid1(n_cur), dim3(256), 0, C->stream, A.tiles, S->L, n_cur, S->opt.max_iter, C->d_order_bins);
      hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_BINS), 0, C->stream, C->d_order_bins);
      hipLaunchKernelGGL(k_order_scatter<T>, grid1(n_cur), dim3(256), 0, C->stream, A.tiles, S->L, n_cur, S->opt.max_iter, C->d_order_bins, C->d_order);
      HIPCHK(hipGetLastError());
      C->order_n = n_cur;
      C->order_epoch = S->inputs_epoch;
    }
    HIPCHK(hipStreamSynchronize(C->stream));
    float ms = 0.f, t0 = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, C->ev_k0, C->ev_k1));
    HIPCHK(hipEventElapsedTime(&t0, S->ev_t0, C->ev_k0));
    if (tail_whole) {
      if (C->arrival_n != n_cur) { C->arrival_n = n_cur; C->arrival_ms = 0.0; }
      if (!tail_ordered) C->arrival_ms = (double)ms;
      else if (tail_order_was_stale && C->arrival_ms > 0.0 && (double)ms > 0.985 * C->arrival_ms) C->order_holdoff = S->tune.flat_order_holdoff;
    }
    C->tail_iv.emplace_back(t0, t0 + ms);
    total_ms += ms;
    iters += C->h_counters[1];
    if (trace)
      fprintf(stderr, "[loikb] tail launch: %6d instances on %u workgroups  %8.3f ms  inst-iters %9u (%.1f M/s)"
                      "  wave-iters %7u H-rebuild %3.0f%%\n",
              n, grid.x, ms, C->h_counters[1], C->h_counters[1] / ms / 1e3,
              C->h_counters[5], 100.0 * C->h_counters[6] / (C->h_counters[5] ? C->h_counters[5] : 1));
    C->stats.launches++;
    C->stats.tail_launches++;
  }
  *ms_out = total_ms;
  *iters_out = iters;
  return LOIKB_OK;
}

// one chunk of the batch, start to finish, on the chunk's stream (called from the chunk's host thread)
template <typename T>
int run_chunk(loikb_solver_impl* S, Chunk* C)
{
  Params<T> P = make_params<T>(S);
  C->stats = loikb_stats{};
  C->unfinished_counted = false;
  C->small_finished = false;
  C->solve_iv.clear();
  C->tail_iv.clear();
  double kernel_ms = 0.0;
  // main-loop bound: at most max_iter-1 iterations, tail solve may reach max_iter (hpp:377, :276)
  const int max_total = S->opt.max_iter + 1;
  const bool can_compact = !(S->opt.flags & LOIKB_OPT_NO_COMPACTION) && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS);
  // compaction pays only while the launch is bandwidth-bound (many wavefronts); below that an ADMM iteration
  // costs the same single-wavefront latency however few lanes are live
  const int compact_min = S->opt.compact_min_instances > 0 ? S->opt.compact_min_instances : 64 * WAVE;
  // k_move costs ~6 KB of traffic per live instance (a fraction of ONE iteration's ~32 KB), so repack eagerly
  const double compact_ratio = S->tune.compact_ratio;
  // cooperative tail kernel (a lane group per instance) once few instances are left
  const bool use_tail = can_compact && S->nb <= WAVE && S->opt.tail_max_instances >= 0;
  // (thresholds are stated for the whole batch: a chunk applies its share)
  const double share = (double)C->B / (double)S->B;
  // (with the lean tail kernel, whole batches up to 2^20 instances go to it directly: it is as fast as the solve kernel's
  //  bulk phase and has neither ragged tiles nor compaction; without it the hand-over is at 32768 live instances)
  const int tail_max = std::max(1, (int)((S->opt.tail_max_instances > 0 ? S->opt.tail_max_instances
                                                                         : S->plan.tail_max) * share));
  const bool trace = S->tune.trace;
  // a team of wavefronts per tile walks independent chains of the tree concurrently: a sweep costs the tree's
  // critical path instead of nb joint visits, and four wavefronts keep four times the loads of a tile in flight.
  // Measured faster than one wavefront per tile at every batch size on a branching robot (Talos: 250 vs 175 M
  // instance-iterations/s in bulk, 30 vs 67 us per iteration for a single tile); pointless on a pure chain.
  const int team_max = S->tune.team_max;
  // LDS of a workgroup: edge slots of the leaf->root sweeps (aliased by the team's scalar exchange) + v slots
  auto edge_entries = [](const loikb_solver_impl::TeamSched& sc) {
    return std::max(std::max(sc.nslots, 1) * EDGE_ENT, sc.nw > 1 ? sc.nw * Norms<T>::NALL : 0);
  };
  auto lds_bytes = [&](const loikb_solver_impl::TeamSched& sc) {
    return (size_t)(edge_entries(sc) + std::max(sc.nvslots, 1) * 6) * WAVE * sizeof(T);
  };
  constexpr size_t LDS_CU = 160 * 1024, LDS_DEFAULT = 64 * 1024;
  const bool team_ok = S->sched[1].nw > 1 && lds_bytes(S->sched[1]) <= LDS_CU &&
                       5 * (S->sched[1].T_up + S->sched[1].T_down) <= 4 * (S->sched[0].T_up + S->sched[0].T_down);
  if (S->plan.solve_ok) {
    const size_t need = std::max(lds_bytes(S->sched[0]), team_ok ? lds_bytes(S->sched[1]) : (size_t)0);
    if (need > LDS_DEFAULT) {
      HIPCHK(hipFuncSetAttribute((const void*)k_solve<T, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      HIPCHK(hipFuncSetAttribute((const void*)k_solve<T, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      HIPCHK(hipFuncSetAttribute((const void*)k_solve<T, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      HIPCHK(hipFuncSetAttribute((const void*)k_solve<T, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    }
  }
  int cur = 0, n_cur = C->B;
  int done_iters = 0;
  unsigned long long inst_iters = 0;
  unsigned int n_live = 0;
  // A batch that is below the hand-over threshold from the start never fills the machine with one instance per lane:
  // it goes to the tail kernel (a lane group per instance, 12 us instead of 30-55 us per iteration) for the whole
  // solve.  (Not when the caller fixed the launch length or asked for the solve kernel's bit-exact behaviour.)
  const bool direct_tail = S->nb <= WAVE && S->opt.tail_max_instances >= 0 && S->opt.max_launch_iters <= 0 &&
                           !(S->opt.flags & (LOIKB_OPT_NO_COMPACTION | LOIKB_OPT_NO_H_CACHE)) && C->B <= tail_max &&
                           S->tune.direct_tail;
  if (!S->plan.solve_ok && !direct_tail) {   // (run_main_loop sends such a solve to k_pass_solve: never here)
    g_last_error = "internal: a tree too bushy for k_solve's LDS slots reached the streaming engine"; return LOIKB_ERR_STATE;
  }
  if (direct_tail) {
    double tms = 0.0;
    unsigned long long tit = 0;
    int rc = run_tail<T>(S, C, P, 0, C->B, C->B, &tms, &tit, true);
    if (rc) return rc;
    kernel_ms += tms;
    if (trace) fprintf(stderr, "[loikb] tail kernel from the first iteration: %d instances  %8.3f ms  inst-iters %9llu\n", C->B, tms, tit);
    C->stats.tail_ms = tms;
    C->stats.tail_instances = C->B;
    C->stats.tail_instance_iterations = tit;
    inst_iters = tit;
  }
  while (!direct_tail) {
    // Latency-bound regime: once the chunk's tiles fit its share of the CUs (one workgroup per CU), an iteration
    // costs the same however few tiles are left -- repacking buys nothing any more and every launch boundary costs a
    // host round trip plus the kernel's prologue/epilogue, so the launches get longer.
    const int tiles_cur = (n_cur + WAVE - 1) / WAVE;
    // (an explicit compact_min_instances is the caller's policy: it is honoured as given)
    const bool latency_bound = S->opt.compact_min_instances <= 0 && tiles_cur * (int)S->chunks.size() <= S->ncu;
    const bool may_compact_later = can_compact && n_cur > compact_min && !latency_bound;
    const int lat_iters = S->tune.lat_iters;
    int launch_iters = S->opt.max_launch_iters > 0 ? S->opt.max_launch_iters
                       : (may_compact_later ? 8 : (use_tail || (can_compact && n_cur > compact_min)) ? lat_iters : max_total);
    if (launch_iters > max_total - done_iters) launch_iters = max_total - done_iters;
    P.B = n_cur;
    P.max_launch_iters = launch_iters;
    Bufs<T> Bf = make_bufs<T>(S, C, cur);
    const loikb_solver_impl::TeamSched& sc = S->sched[(team_ok && n_cur <= team_max) ? 1 : 0];
    C->stats.team = sc.nw;
    const int edge_ent = edge_entries(sc);
    const Team tm{sc.d_up, sc.d_down, sc.d_rlist, sc.T_up, sc.T_down, edge_ent};
    const size_t lds = lds_bytes(sc);
    const dim3 grid((unsigned)((n_cur + WAVE - 1) / WAVE)), block(WAVE * sc.nw);
    HIPCHK(hipMemsetAsync(C->d_counters, 0, NCOUNTERS * sizeof(unsigned int), C->stream));
    HIPCHK(hipEventRecord(C->ev_k0, C->stream));
    if (sc.nw > 1) {
      if (S->href_diag) hipLaunchKernelGGL((k_solve<T, true, true>), grid, block, lds, C->stream, P, Bf, tm.up, tm.down, tm.rlist, tm.T_up, tm.T_down, tm.edge_ent);
      else hipLaunchKernelGGL((k_solve<T, false, true>), grid, block, lds, C->stream, P, Bf, tm.up, tm.down, tm.rlist, tm.T_up, tm.T_down, tm.edge_ent);
    } else {
      if (S->href_diag) hipLaunchKernelGGL((k_solve<T, true, false>), grid, block, lds, C->stream, P, Bf, tm.up, tm.down, tm.rlist, tm.T_up, tm.T_down, tm.edge_ent);
      else hipLaunchKernelGGL((k_solve<T, false, false>), grid, block, lds, C->stream, P, Bf, tm.up, tm.down, tm.rlist, tm.T_up, tm.T_down, tm.edge_ent);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(C->ev_k1, C->stream));
    HIPCHK(hipMemcpyAsync(C->h_counters, C->d_counters, NCOUNTERS * sizeof(unsigned int), hipMemcpyDeviceToHost, C->stream));
    HIPCHK(hipStreamSynchronize(C->stream));
    float ms = 0.f, t0 = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, C->ev_k0, C->ev_k1));
    HIPCHK(hipEventElapsedTime(&t0, S->ev_t0, C->ev_k0));
    C->solve_iv.emplace_back(t0, t0 + ms);
    kernel_ms += ms;
    C->stats.launches++;
    inst_iters += C->h_counters[1];
    n_live = C->h_counters[0];
    done_iters += launch_iters;
    if (trace)
      fprintf(stderr, "[loikb] launch %3d: set %d slots %7d iters %4d..%4d  %8.3f ms  inst-iters %9u (%.1f M/s)  live after %7u"
                      "  tile-iters %6u H-rebuild %3.0f%% fused %3.0f%%\n",
              C->stats.launches, cur, n_cur, done_iters - launch_iters + 1, done_iters, ms, C->h_counters[1],
              C->h_counters[1] / ms / 1e3, n_live, C->h_counters[2],
              100.0 * C->h_counters[3] / (C->h_counters[2] ? C->h_counters[2] : 1),
              100.0 * C->h_counters[4] / (C->h_counters[2] ? C->h_counters[2] : 1));
    if (n_live == 0 || done_iters >= max_total) break;
    if (use_tail && (int)n_live <= tail_max) {
      double tms = 0.0;
      unsigned long long tit = 0;
      int rc = run_tail<T>(S, C, P, cur, n_cur, (int)n_live, &tms, &tit);
      if (rc) return rc;
      kernel_ms += tms;
      if (trace) fprintf(stderr, "[loikb] tail kernel: %u instances  %8.3f ms  inst-iters %9llu\n", n_live, tms, tit);
      C->stats.tail_ms = tms;
      C->stats.tail_instances = (int)n_live;
      C->stats.tail_instance_iterations = tit;
      inst_iters += tit;
      n_live = 0;
      break;
    }
    if (may_compact_later && (double)n_live <= compact_ratio * n_cur) {
      const int dst = cur == 1 ? 2 : 1;
      int n_new = 0;
      int rc = compact<T>(S, C, cur, dst, n_cur, &n_new);
      if (rc) return rc;
      cur = dst;
      n_cur = n_new;
      C->stats.compactions++;
    }
  }
  if (cur != 0) {
    int rc = compact<T>(S, C, cur, -1, n_cur, nullptr);  // everything that is still in a work set goes home
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(C->stream));
  C->stats.instance_iterations = inst_iters;
  if (!C->unfinished_counted) C->stats.n_unfinished = (int)n_live;
  C->stats.kernel_ms = kernel_ms;
  return LOIKB_OK;
}

// fork: one host thread + stream per chunk; join: sum the per-chunk statistics
template <typename T>
int run_main_loop_t(loikb_solver_impl* S)
{
  S->stats = loikb_stats{};
  ++S->solve_serial;
  S->stats.bytes_per_instance_iteration = (double)sizeof(T) * (203.0 * S->nb + 108.0 * S->nc);
  HIPCHK(hipEventRecord(S->ev_t0, S->stream));
  const int nchunks = (int)S->chunks.size();
  if (nchunks == 1) {
    Chunk* C = &S->chunks[0];
    C->stream = S->stream;
    int rc = run_chunk<T>(S, C);
    if (rc) return rc;
  } else {
    // the chunk streams start after everything queued on the caller's stream (SolveInit uploads, resets)
    HIPCHK(hipEventRecord(S->ev_fork, S->stream));
    for (Chunk& C : S->chunks) HIPCHK(hipStreamWaitEvent(C.stream, S->ev_fork, 0));
    // (Starting chunk k+1 only when chunk k reaches its straggler phase was measured too: 61 vs 48 ms/step -- two
    // bulk phases side by side cost less than a bulk phase squeezed between tail-kernel workgroups.)
    std::vector<std::thread> th;
    for (Chunk& C : S->chunks)
      th.emplace_back([S, &C]() {
        if (hipSetDevice(S->device) != hipSuccess) { C.rc = LOIKB_ERR_HIP; C.err = "hipSetDevice failed in a chunk thread"; return; }
        C.rc = run_chunk<T>(S, &C);
        if (C.rc) C.err = g_last_error;
      });
    for (auto& t : th) t.join();
    for (Chunk& C : S->chunks)
      if (C.rc) { g_last_error = C.err; return C.rc; }
  }
  // n_unfinished: instances that stopped at max_iter, neither converged nor flagged (whatever engine finished them)
  Chunk* C0 = &S->chunks[0];
  const bool counted = nchunks == 1 && C0->unfinished_counted;   // (a small batch: the on-chip launch's own counters said it)
  if (!counted) {
    HIPCHK(hipMemsetAsync(C0->d_counters, 0, sizeof(unsigned int), S->stream));
    hipLaunchKernelGGL(k_count_unfinished<T>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, S->B, C0->d_counters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(C0->h_counters, C0->d_counters, sizeof(unsigned int), hipMemcpyDeviceToHost, S->stream));
  }
  float tms = 0.f;
  if (counted && C0->small_finished) {
    // (the short sequence: run_tail has waited for its last kernel, which left the counters on the host; nothing was queued since --
    //  no event and no synchronisation of this function's own)
    HIPCHK(hipEventElapsedTime(&tms, S->ev_t0, C0->ev_k1));
  } else {
    HIPCHK(hipEventRecord(S->ev_t1, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    HIPCHK(hipEventElapsedTime(&tms, S->ev_t0, S->ev_t1));
  }
  S->stats.n_unfinished = counted ? C0->stats.n_unfinished : (int)C0->h_counters[0];
  for (const Chunk& C : S->chunks) {
    S->stats.instance_iterations += C.stats.instance_iterations;
    S->stats.launches += C.stats.launches;
    S->stats.compactions += C.stats.compactions;
    S->stats.tail_instances += C.stats.tail_instances;
    S->stats.tail_ms += C.stats.tail_ms;
    S->stats.kernel_ms += C.stats.kernel_ms;
    S->stats.tail_instance_iterations += C.stats.tail_instance_iterations;
    S->stats.tail_launches += C.stats.tail_launches;
    S->stats.lean_launches += C.stats.lean_launches;
    S->stats.flat_launches += C.stats.flat_launches;
    S->stats.flat_split_launches += C.stats.flat_split_launches;
    S->stats.flat_ordered += C.stats.flat_ordered;
    S->stats.flat_built += C.stats.flat_built;
    S->stats.flat_probe_launches += C.stats.flat_probe_launches;
    S->stats.probe_ms += C.stats.probe_ms;
    S->stats.queue_dry_ms += C.stats.queue_dry_ms;
    S->stats.lean_escaped += C.stats.lean_escaped;
    S->stats.hslots_ms += C.stats.hslots_ms;
    S->stats.lean_requeues += C.stats.lean_requeues;
    S->stats.team = C.stats.team;
  }
  S->stats.chunks = nchunks;
  S->stats.total_ms = tms;
  // time during which at least one launch of a kernel was executing (= the sum of its launch times when chunks == 1)
  auto busy = [&](bool tail) {
    std::vector<std::pair<float, float>> iv;
    for (const Chunk& C : S->chunks) {
      const auto& v = tail ? C.tail_iv : C.solve_iv;
      iv.insert(iv.end(), v.begin(), v.end());
    }
    std::sort(iv.begin(), iv.end());
    double tot = 0.0;
    float lo = 0.f, hi = -1.f;
    for (const auto& x : iv) {
      if (hi < lo || x.first > hi) { if (hi >= lo) tot += hi - lo; lo = x.first; hi = x.second; }
      else if (x.second > hi) hi = x.second;
    }
    if (hi >= lo) tot += hi - lo;
    return tot;
  };
  S->stats.solve_busy_ms = busy(false);
  S->stats.tail_busy_ms = busy(true);
  return LOIKB_OK;
}


int run_pass_solve(loikb_solver_impl* S, bool logged);   // (the plain pass-by-pass implementation: defined beside run_logged)
// LOIKB_MU_MAXEIGENVALUE: the solve's starting mu, from the references in force now (SolveInit's reset ran before they were stored)
static int start_mu(loikb_solver_impl* S)
{
  if (S->opt.mu_update_strat != LOIKB_MU_MAXEIGENVALUE) return LOIKB_OK;
  const double mu = spectral_mu0(S);
  if (mu != S->mu_start) { S->seen_lo = 1 << 20; S->seen_hi = -(1 << 20); S->end_hist_n = 0; }  // the decades are counted from this mu
  S->mu_start = mu;
  return reset_home(S, RS_MU);
}

int run_main_loop(loikb_solver_impl* S)
{
  // (reset_home's offer of a prepared queue holds for the solve it was made for, however this function is left)
  struct QueueOffer { loikb_solver_impl* S; ~QueueOffer() { for (loikb_solver_impl::Chunk& C : S->chunks) C.queue_ready = false; } } offer{S};
  S->pass_active = false;  // (pass-level calls work on a copy of the state: a solve continues from the solver's own)
  // UpdateMu's throw site for an unknown strategy (hxx:638-640); OSQP and MAXEIGENVALUE are extensions of this library
  if (S->opt.mu_update_strat != LOIKB_MU_DEFAULT && S->opt.mu_update_strat != LOIKB_MU_OSQP &&
      S->opt.mu_update_strat != LOIKB_MU_MAXEIGENVALUE && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS)) {
    g_last_error = "[FirstOrderLoikOptimizedTpl::UpdateMu]: mu update strategy not supported";
    return LOIKB_ERR_MU_STRATEGY;
  }
  int rc;
  if (!S->plan.solve_ok && !bushy_goes_on_chip(S)) rc = run_pass_solve(S, false);   // (the engine of last resort: EnginePlan::solve_ok)
  else if ((rc = start_mu(S))) {}
  else rc = S->f32 ? run_main_loop_t<float>(S) : run_main_loop_t<double>(S);
  return rc;
}

// liMi of the caller's joints: sel[e] + 1 = the first device joint of joint e + 1, last[e] + 1 = the last one.  One device
// joint carries the whole M(q) for 1-DoF joints and for the chains about ONE frame (free-flyer, spherical, translation,
// planar: the other chain joints are the identity); a SphericalZYX joint is the product of its three revolute joints.
template <typename T>
__global__ void k_limi(char* tiles, Layout L, const JointDesc* __restrict__ jd, const int* __restrict__ sel,
                       const int* __restrict__ last, int nsel, int B, double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  for (int e = 0; e < nsel; ++e) {
    double Ra[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ta[3] = {0, 0, 0};
    for (int i = sel[e] + 1; i <= last[e] + 1; ++i) {
      T R[9], t[3];
      const char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
      const typename Vec2<T>::type cs = ldp<T>(rec, JP_CS);
      joint_xform<T>(jd[i], rec, cs.x, cs.y, R, t);
      double Rn[9], tn[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rn[3 * r + c] = Ra[3 * r] * (double)R[c] + Ra[3 * r + 1] * (double)R[3 + c] + Ra[3 * r + 2] * (double)R[6 + c];
        tn[r] = ta[r] + Ra[3 * r] * (double)t[0] + Ra[3 * r + 1] * (double)t[1] + Ra[3 * r + 2] * (double)t[2];
      }
      for (int k = 0; k < 9; ++k) Ra[k] = Rn[k];
      for (int k = 0; k < 3; ++k) ta[k] = tn[k];
    }
    double* o = out + ((size_t)b * nsel + e) * 12;
    for (int k = 0; k < 9; ++k) o[k] = Ra[k];
    for (int k = 0; k < 3; ++k) o[9 + k] = ta[k];
  }
}

// get_primal_residual_vec() / get_dual_residual_vec() (loik-loid-optimized.hpp:698-699), [6 nl + nv] per instance, rebuilt
// from the state of the last iteration -- the hot path only ever forms their running maxima:
//   primal: rows 6(c_id-1).. = A_c v_c - b_c for the constrained links, 0 elsewhere (hxx:433);
//           tail = nu - z (hxx:394)
//   dual:   rows 6(i-1).. = H_ref v_i - H_ref v_ref + g_i (hxx:228);  tail = S^T f + w (hxx:484)
template <typename T>
__global__ void k_residual_vecs(char* tiles, Layout L, const JointDesc* __restrict__ jd, const T* __restrict__ uni,
                                const double* __restrict__ href_tab, int a_shared, const int* __restrict__ sel, int nl, int B, int dual,
                                double* __restrict__ out)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  char* lp = lane_ptr<T>(tiles, L, b);
  double* o = out + (size_t)b * (6 * nl + L.nb);
  for (int e = 0; e < nl; ++e) {
    const int i = sel[e] + 1;
    const char* rec = lp + (size_t)(i - 1) * JREC * pair_bytes<T>();
    T v[6];
    ld6<T>(rec, JP_V, v);
    if (dual) {
      T g[6];
      ld6<T>(rec, JP_G, g);
      const double* row = href_tab + (size_t)i * HREF_ROW;
      for (int r = 0; r < 6; ++r) {
        double a = 0.0;
        for (int k = 0; k < 6; ++k) a += row[6 * r + k] * (double)v[k];
        o[6 * e + r] = a - row[36 + r] + (double)g[r];
      }
    } else {
      const int cs = jd[i].cslot;
      for (int r = 0; r < 6; ++r) o[6 * e + r] = 0.0;
      if (cs >= 0) {
        const char* crec = lp + (size_t)(L.off_c + cs * L.crec) * pair_bytes<T>();
        T bb[6];
        ld6<T>(crec, CP_B, bb);
        for (int r = 0; r < 6; ++r) {
          double a = 0.0;
          for (int k = 0; k < 6; ++k) {
            const int q = 6 * r + k;
            const T A = a_shared ? uni[cs * 36 + q] : *elem_ptr<T>(const_cast<char*>(crec), CP_A + q / 2, q & 1);
            a += (double)A * (double)v[k];
          }
          o[6 * e + r] = a - (double)bb[r];
        }
      }
    }
  }
  for (int j = 0; j < L.nb; ++j) {
    const char* rec = lp + (size_t)j * JREC * pair_bytes<T>();
    const typename Vec2<T>::type nus = ldp<T>(rec, JP_NUS);
    o[6 * nl + j] = dual ? (double)nus.y : (double)nus.x - (double)ldp<T>(rec, JP_WZ).y;
  }
}

// dst[b][e][:] = src[b][sel[e]][:]  (rows of w doubles): the bodies of the caller's model out of the device tree's
__global__ void k_select_rows(const double* __restrict__ src, int nrow_src, int w, const int* __restrict__ sel, int nsel,
                              int B, double* __restrict__ dst)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int e = 0; e < nsel; ++e)
    for (int k = 0; k < w; ++k) dst[((size_t)b * nsel + e) * w + k] = src[((size_t)b * nrow_src + sel[e]) * w + k];
}

}  // namespace

struct loikb_solver : loikb_solver_impl {};

extern "C" {

int loikb_version(void) { return LOIKB_VERSION; }

int loikb_sweep_schedule(const int* parents, int njoints, int team, int direction, int steps_cap, int* joint_out,
                         int* flags_out, int* slot_out, int* lds_slots_out)
{
  if (!parents || njoints < 2 || team < 1 || team > MAX_TEAM || (direction != 0 && direction != 1) || !joint_out)
    return LOIKB_ERR_ARG;
  for (int i = 1; i < njoints; ++i)
    if (parents[i] < 0 || parents[i] >= i) return LOIKB_ERR_ARG;
  loikb_solver_impl::TeamSched sc;
  build_team_schedule(std::vector<int>(parents, parents + njoints), team, sc);
  const int T = direction == 0 ? sc.T_up : sc.T_down;
  const std::vector<StepDesc>& st = direction == 0 ? sc.up : sc.down;
  if (T > steps_cap) return LOIKB_ERR_ARG;
  for (int w = 0; w < team; ++w)
    for (int t = 0; t < steps_cap; ++t) {
      const bool in = t < T;
      joint_out[w * steps_cap + t] = in ? st[(size_t)w * T + t].joint : 0;
      if (flags_out) flags_out[w * steps_cap + t] = in ? st[(size_t)w * T + t].flags : 0;
      if (slot_out) slot_out[w * steps_cap + t] = in && (st[(size_t)w * T + t].flags & SF_OUT_LDS) ? st[(size_t)w * T + t].wslot : -1;
    }
  if (lds_slots_out) *lds_slots_out = direction == 0 ? sc.nslots : sc.nvslots;
  return T;
}

// the flat engine's static schedule of a tree (inspection / tests, like loikb_sweep_schedule)
int loikb_flat_schedule(const int* parents, int njoints, int* out, int cap, int* meta)
{
  if (!parents || njoints < 2 || !meta) return LOIKB_ERR_ARG;
  for (int i = 1; i < njoints; ++i)
    if (parents[i] < 0 || parents[i] >= i) { g_last_error = "parents[i] must be < i"; return LOIKB_ERR_MODEL; }
  loikb_solver_impl::FlatSched fs;
  build_flat_schedule(std::vector<int>(parents, parents + njoints), fs);
  meta[0] = fs.ok; meta[1] = fs.G; meta[2] = fs.nanc; meta[3] = fs.nscan; meta[4] = fs.njmp;
  if (!fs.ok) { g_last_error = fs.why; return LOIKB_OK; }
  constexpr int W = (int)(sizeof(FlatLane) / sizeof(int));
  static_assert(sizeof(FlatLane) == sizeof(int) * (2 + FLAT_JMP + FLAT_MAXA + FLAT_RED + 1 + FLAT_PART), "FlatLane is a plain int record");
  if (!out || cap < fs.G * W) return fs.G * W;
  memcpy(out, fs.lanes.data(), sizeof(FlatLane) * fs.lanes.size());
  for (int l = 0; l < fs.G; ++l) out[l * W + (int)(offsetof(FlatLane, helper) / sizeof(int))] &= 1;  // (the upper bits are internal)
  return LOIKB_OK;
}

const char* loikb_last_error(void) { return g_last_error.c_str(); }

const char* loikb_status_string(int code)
{
  switch (code) {
  case LOIKB_OK: return "ok";
  case LOIKB_ERR_EQ_C_DIM:
    return "[IkProblemFormulation::IkProblemFormulation]: equality constraint dimension is not 6, problem formulation "
           "not supported !!!";
  case LOIKB_ERR_EQ_C_SIZE:
    return "[IkProblemFormulation::UpdateEqConstraints]: number of equality constraints doesn't match initialization!!!";
  case LOIKB_ERR_INEQ_DIM:
    return "IkProblemFormulation::UpdateIneqConstraints]: inequality constraint dimension has changed, this is not "
           "supported currently!!!";
  case LOIKB_ERR_NO_SUCH_CONSTRAINT:
    return "[IkProblemFormulation::UpdateEqConstraint]: constraint doesn't yet exist at link 'c_id' !!! ";
  case LOIKB_ERR_DUP_CONSTRAINT:
    return "[IkProblemFormulation::UpdateEqConstraint]: multiple constraint specification for the same link id, not "
           "supported, terminating !!!";
  case LOIKB_ERR_MU_STRATEGY: return "[FirstOrderLoikOptimizedTpl::UpdateMu]: mu update strategy not supported";
  case LOIKB_ERR_MODEL:
    return "[IkProblemFormulation::IkProblemFormulation]: nb does not equal to nj - 1, robot model not supported !!!";
  case LOIKB_ERR_REFS_SIZE:
    return "[IkProblemFormulation::UpdateReferences]: input arguments 'H_refs', 'v_refs' have wrong size!!";
  case LOIKB_ERR_ARG: return "invalid argument";
  case LOIKB_ERR_HIP: return "HIP runtime error";
  case LOIKB_ERR_NO_DEVICE: return "no HIP device";
  case LOIKB_ERR_HREF_NOT_SYMMETRIC: return "H_ref must be symmetric";
  case LOIKB_ERR_STATE: return "Solve() called before SolveInit()";
  default: return "unknown";
  }
}

int loikb_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int loikb_create(const loikb_model_desc* model, const loikb_options* opts, loikb_solver** out)
{
  if (!model || !opts || !out) return LOIKB_ERR_ARG;
  if (opts->eq_c_dim != 6) return LOIKB_ERR_EQ_C_DIM;
  if (opts->batch < 1) { g_last_error = "batch must be >= 1"; return LOIKB_ERR_ARG; }
  if (opts->num_eq_c < 0) { g_last_error = "num_eq_c must be >= 0"; return LOIKB_ERR_ARG; }
  if (opts->eq_c_capacity < 0) { g_last_error = "eq_c_capacity must be >= 0"; return LOIKB_ERR_ARG; }
  // one constraint per link at most (hpp:197-199) and every slot sits on a body: no more slots than bodies
  if (std::max(opts->num_eq_c, opts->eq_c_capacity) > model->njoints - 1) {
    g_last_error = "more constraint slots (num_eq_c / eq_c_capacity) than bodies";
    return LOIKB_ERR_EQ_C_SIZE;
  }
  loikb_solver* S = new loikb_solver();
  S->tune.read_env();
  int rc = build_schedule(S, model);
  if (rc) { delete S; return rc; }
  build_flat_schedule(S->parents, S->flat);
  S->opt = *opts;
  S->B = opts->batch;
  S->nc = std::max(opts->num_eq_c, opts->eq_c_capacity);
  S->nc_active = opts->num_eq_c;
  S->f32 = opts->precision == LOIKB_F32;
  S->esz = S->f32 ? 4 : 8;
  S->device = opts->device;
  if (loikb_device_count() <= S->device) {
    g_last_error = "no HIP device available for the requested ordinal";
    delete S;
    return LOIKB_ERR_NO_DEVICE;
  }
  auto fail = [&](int code) { loikb_destroy(S); return code; };
#define TRY(x) do { int _rc = (x); if (_rc) return fail(_rc); } while (0)
#define HIPTRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) { g_last_error = std::string(#x) + ": " + hipGetErrorString(_e); return fail(LOIKB_ERR_HIP); } } while (0)
  HIPTRY(hipSetDevice(S->device));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, S->device) == hipSuccess && prop.multiProcessorCount > 0) S->ncu = prop.multiProcessorCount;
  }
  if (opts->flags & LOIKB_OPT_OWN_STREAM) {
    HIPTRY(hipStreamCreateWithFlags(&S->own_stream, hipStreamNonBlocking));
    S->stream = S->own_stream;
  }
  HIPTRY(hipEventCreate(&S->ev_t0));
  HIPTRY(hipEventCreate(&S->ev_t1));
  HIPTRY(hipEventCreate(&S->ev_fork));
  void* tmp = nullptr;
  TRY(alloc_dev(S, &tmp, sizeof(JointDesc) * S->nj)); S->d_jd = (JointDesc*)tmp;
  TRY(alloc_dev(S, &tmp, sizeof(int) * S->nj)); S->d_idx_q = (int*)tmp;
  TRY(alloc_dev(S, &tmp, sizeof(int) * ROWMAP_CAP)); S->d_rowmap = (int*)tmp;
  TRY(alloc_dev(S, &tmp, sizeof(double) * (size_t)S->B * S->nq)); S->d_q = (double*)tmp;
  TRY(alloc_dev(S, &tmp, sizeof(TailTopo) * S->nj)); S->d_topo = (TailTopo*)tmp;
  TRY(alloc_dev(S, &tmp, sizeof(int) * (S->child_list.size() + 1))); S->d_child_list = (int*)tmp;
  HIPTRY(hipMemcpyAsync(S->d_topo, S->topo.data(), sizeof(TailTopo) * S->nj, hipMemcpyHostToDevice, S->stream));
  if (!S->child_list.empty())
    HIPTRY(hipMemcpyAsync(S->d_child_list, S->child_list.data(), sizeof(int) * S->child_list.size(),
                          hipMemcpyHostToDevice, S->stream));
  TRY(alloc_dev(S, &S->d_uni, S->esz * ((size_t)(S->nc > 0 ? S->nc : 1) * 57 + 2 * (size_t)S->nb)));
  HIPTRY(hipMemcpyAsync(S->d_idx_q, S->idx_q.data(), sizeof(int) * S->nj, hipMemcpyHostToDevice, S->stream));
  {
    std::vector<int> sel(S->ext_nj), fsel(S->ext_nj);
    for (int i = 1; i < S->ext_nj; ++i) { sel[i - 1] = S->link_of[i] - 1; fsel[i - 1] = S->first_of[i] - 1; }
    TRY(alloc_dev(S, &tmp, sizeof(int) * S->ext_nj)); S->d_link_sel = (int*)tmp;
    TRY(alloc_dev(S, &tmp, sizeof(int) * S->ext_nj)); S->d_first_sel = (int*)tmp;
    HIPTRY(hipMemcpyAsync(S->d_link_sel, sel.data(), sizeof(int) * (S->ext_nj - 1), hipMemcpyHostToDevice, S->stream));
    HIPTRY(hipMemcpyAsync(S->d_first_sel, fsel.data(), sizeof(int) * (S->ext_nj - 1), hipMemcpyHostToDevice, S->stream));
    HIPTRY(hipStreamSynchronize(S->stream));  // the host vectors go out of scope
  }
  for (auto& sc : S->sched) {
    TRY(alloc_dev(S, &tmp, sizeof(StepDesc) * sc.up.size())); sc.d_up = (StepDesc*)tmp;
    TRY(alloc_dev(S, &tmp, sizeof(StepDesc) * sc.down.size())); sc.d_down = (StepDesc*)tmp;
    TRY(alloc_dev(S, &tmp, sizeof(int) * sc.rlist.size())); sc.d_rlist = (int*)tmp;
    HIPTRY(hipMemcpyAsync(sc.d_rlist, sc.rlist.data(), sizeof(int) * sc.rlist.size(), hipMemcpyHostToDevice, S->stream));
  }
  if (S->flat.ok) {
    TRY(alloc_dev(S, &tmp, sizeof(FlatLane) * S->flat.lanes.size())); S->flat.d_lanes = (FlatLane*)tmp;
    HIPTRY(hipMemcpyAsync(S->flat.d_lanes, S->flat.lanes.data(), sizeof(FlatLane) * S->flat.lanes.size(), hipMemcpyHostToDevice, S->stream));
  }
  TRY(upload_jd(S));
  TRY(ensure_layout(S, true));
  HIPTRY(hipStreamSynchronize(S->stream));
#undef TRY
#undef HIPTRY
  *out = S;
  return LOIKB_OK;
}

int loikb_destroy(loikb_solver* S)
{
  if (!S) return LOIKB_OK;
  (void)hipSetDevice(S->device);
  destroy_chunks(S);  // (first: it takes its buffers out of `allocs`)
  for (void* a : S->allocs) if (a) (void)hipFree(a);
  if (S->d_stage) (void)hipFree(S->d_stage);
  for (int k = 0; k < 2; ++k) if (S->d_getscr[k]) (void)hipFree(S->d_getscr[k]);
  for (auto& rmap : S->resmaps) if (rmap.d) (void)hipFree(rmap.d);
  if (S->h_pin) (void)hipHostFree(S->h_pin);
  if (S->h_res) (void)hipHostFree(S->h_res);
  if (S->d_pass) (void)hipFree(S->d_pass);
  if (S->d_pass_cslot) (void)hipFree(S->d_pass_cslot);
  if (S->d_log) (void)hipFree(S->d_log);
  if (S->d_log_rows) (void)hipFree(S->d_log_rows);
  (void)hipGetLastError();  // a failed free must not surface in the next solver's first launch check
  if (S->own_stream) (void)hipStreamDestroy(S->own_stream);
  if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
  if (S->ev_t0) (void)hipEventDestroy(S->ev_t0);
  if (S->ev_t1) (void)hipEventDestroy(S->ev_t1);
  delete S;
  return LOIKB_OK;
}

int loikb_set_stream(loikb_solver* S, void* hip_stream)
{
  if (!S) return LOIKB_ERR_ARG;
  S->stream = (hipStream_t)hip_stream;
  return LOIKB_OK;
}

// final_sync = false (loikb_solve_full): the solve that follows ends with a synchronisation of its own -- the caller's arrays are read by then
static int solve_init_impl(loikb_solver* S, const double* q, const double* H_ref, const double* v_ref, const int* c_ids,
                           int nc, const double* Ais, const double* bis, const double* lb, const double* ub, int nbound,
                           int in_flags, bool final_sync)
{
  if (!S || !q || !H_ref || !v_ref || (nc > 0 && (!c_ids || !Ais || !bis)) || !lb || !ub) return LOIKB_ERR_ARG;
  HIPCHK(hipSetDevice(S->device));
  int rc;
  if ((rc = validate_problem(S, H_ref, c_ids, nc, nbound))) return rc;   // (nothing of the handle has changed yet)
  memcpy(S->Href, H_ref, 36 * sizeof(double));   // (the plan looks at the reference weight: set_problem stores it again)
  S->href_known = true;
  S->per_link = false;   // (UpdateReference replaces a per-link table: the plan and the slot buffers are sized for the problem being set)
  if ((rc = ensure_layout(S, in_flags & LOIKB_A_SHARED))) return rc;
  S->pass_active = false;
  // (the uploads queue up behind each other; the caller's arrays are read by the time this function returns: ONE synchronisation, below)
  struct Deferred { loikb_solver_impl* S; bool armed; ~Deferred() { if (armed) (void)flush_uni(S); S->defer_sync = false; if (armed) (void)hipStreamSynchronize(S->stream); } } deferred{S, true};
  S->defer_sync = true;
  S->pin_off = 0;
  // problem_.Reset(); ik_id_data_.Reset(warm_start); ResetSolver()  (hpp:345-352)
  if ((rc = reset_home(S, RS_SOLVER | (S->opt.warm_start ? 0 : RS_DATA_COLD)))) return rc;
  if ((rc = set_problem(S, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, nbound, in_flags))) return rc;
  if ((rc = fwd_pass_init(S, q, in_flags))) return rc;
  if ((rc = flush_uni(S))) return rc;
  S->defer_sync = false;
  deferred.armed = false;
  if (final_sync) HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

int loikb_solve_init(loikb_solver* S, const double* q, const double* H_ref, const double* v_ref, const int* c_ids,
                     int nc, const double* Ais, const double* bis, const double* lb, const double* ub, int nbound,
                     int in_flags)
{
  return solve_init_impl(S, q, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, nbound, in_flags, true);
}

// problem_.UpdateReferences(H_refs, v_refs), ik-id-description-optimized.hpp:103-121
int loikb_update_references(loikb_solver* S, const double* H_refs, const double* v_refs, int n)
{
  if (!S || !H_refs || !v_refs) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "UpdateReferences() before SolveInit()"; return LOIKB_ERR_STATE; }
  if (n != S->ext_nj) { g_last_error = "'H_refs', 'v_refs' have wrong size (one entry per joint incl. the universe)"; return LOIKB_ERR_REFS_SIZE; }
  for (int e = 1; e < n; ++e)
    if (!symmetric6(H_refs + 36 * e)) { g_last_error = "H_refs[i] must be symmetric"; return LOIKB_ERR_HREF_NOT_SYMMETRIC; }
  HIPCHK(hipSetDevice(S->device));
  HIPCHK(hipStreamSynchronize(S->stream));
  ++S->inputs_epoch;
  fill_href_tab(S, H_refs, v_refs, 36, 6);
  S->tab_bcast = false;
  // Hv_inf_norm_ is not reset here and the universe's entry counts (hpp:110-118: the loop runs over all nj entries)
  S->href_diag = true;
  for (int e = 0; e < n; ++e) {
    const double *He = H_refs + 36 * e, *ve = v_refs + 6 * e;
    for (int i = 0; i < 6; ++i) {
      double a = 0.0;
      for (int k = 0; k < 6; ++k) {
        a += He[6 * i + k] * ve[k];
        if (e > 0 && i != k && He[6 * i + k] != 0.0) S->href_diag = false;
      }
      if (std::fabs(a) > S->Hv_inf_norm) S->Hv_inf_norm = std::fabs(a);
    }
  }
  S->per_link = true;
  S->pass_active = false;   // (the pass-level path re-reads the problem on its next call)
  int rc;
  if ((rc = upload_href_tab(S))) return rc;
  if ((rc = ensure_hslots(S))) return rc;
  return reset_home(S, RS_HCACHE);  // H_i = rho I + H_ref_i + ...: the cached factors are stale
}

static int run_logged(loikb_solver_impl* S, int redo_reset = 0);

int loikb_solve(loikb_solver* S)
{
  if (!S) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "Solve() before SolveInit()"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  int rc;
  // ik_id_data_.ResetRecursion(); ResetSolver()  (hpp:370-374)
  if ((rc = reset_home(S, RS_RECURSION | RS_SOLVER, !S->opt.logging))) return rc;
  return S->opt.logging ? run_logged(S, RS_RECURSION | RS_SOLVER) : run_main_loop(S);
}

int loikb_solve_full(loikb_solver* S, const double* q, const double* H_ref, const double* v_ref, const int* c_ids,
                     int nc, const double* Ais, const double* bis, const double* lb, const double* ub, int nbound,
                     int in_flags)
{
  if (!S) return LOIKB_ERR_ARG;
  S->offer_queue = !S->opt.logging;
  int rc = solve_init_impl(S, q, H_ref, v_ref, c_ids, nc, Ais, bis, lb, ub, nbound, in_flags, false);
  S->offer_queue = false;
  if (rc) { for (loikb_solver_impl::Chunk& C : S->chunks) C.queue_ready = false; return rc; }
  rc = S->opt.logging ? run_logged(S, S->opt.warm_start ? 0 : (RS_SOLVER | RS_DATA_COLD | RS_Y | RS_HCACHE)) : run_main_loop(S);
  if (rc) (void)hipStreamSynchronize(S->stream);   // (a solve that fails may return before its own: the caller's arrays must have been read)
  return rc;
}

int loikb_solve_tailored(loikb_solver* S, const double* q, int c_id, const double* Ai, const double* bi, int in_flags)
{
  if (!S || (c_id >= 0 && (!Ai || !bi))) return LOIKB_ERR_ARG;  // q == NULL: the configurations resident on the device
  if (!S->have_problem) { g_last_error = "tailored Solve() before SolveInit()"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  int rc;
  // (as in SolveInit: the uploads queue up; whichever way this function is left, the caller's arrays have been read)
  struct Deferred { loikb_solver_impl* S; bool armed; ~Deferred() { if (armed) (void)flush_uni(S); S->defer_sync = false; if (armed) (void)hipStreamSynchronize(S->stream); } } deferred{S, true};
  S->defer_sync = true;
  S->pin_off = 0;
  // ik_id_data_.Reset(warm_start); ResetSolver()  (hpp:604-608)
  if ((rc = reset_home(S, RS_SOLVER | (S->opt.warm_start ? 0 : RS_DATA_COLD)))) return rc;
  // problem_.UpdateEqConstraint(c_id, Ai, bi), ik-id-description-optimized.hpp:178-218.  c_id < 0: no constraint update (not
  // upstream: the way to solve after AddEqConstraint / RemoveEqConstraint changed the set, possibly to the empty one)
  if (c_id >= 0 && (rc = update_eq_single(S, c_id, Ai, bi, in_flags))) return rc;
  S->offer_queue = !S->opt.logging;
  rc = fwd_pass_init(S, q, in_flags);
  S->offer_queue = false;
  if (rc == LOIKB_OK) rc = flush_uni(S);
  if (rc) { for (loikb_solver_impl::Chunk& C : S->chunks) C.queue_ready = false; return rc; }
  S->defer_sync = false;
  deferred.armed = false;   // (the solve below ends with its own synchronisation)
  return S->opt.logging ? run_logged(S, S->opt.warm_start ? 0 : (RS_SOLVER | RS_DATA_COLD | RS_Y | RS_HCACHE)) : run_main_loop(S);
}

// problem_.UpdateEqConstraint (hpp:178-238), AddEqConstraint (:244-286), RemoveEqConstraint (:292-319) between solves
int loikb_update_eq_constraint(loikb_solver* S, int c_id, const double* Ai, const double* bi, int in_flags)
{
  if (!S || !bi) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "UpdateEqConstraint() before SolveInit()"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  int rc;
  if ((rc = update_eq_single(S, c_id, Ai, bi, in_flags))) return rc;
  S->pass_active = false;
  return reset_home(S, RS_HCACHE);  // H_i = rho I + H_ref + mu_eq AtA: the cached factors are stale
}

int loikb_add_eq_constraint(loikb_solver* S, int c_id, const double* Ai, const double* bi, int in_flags)
{
  if (!S || !Ai || !bi) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "AddEqConstraint() before SolveInit()"; return LOIKB_ERR_STATE; }
  for (int c = 0; c < S->nc_active; ++c)
    if (S->active_ids[c] == c_id) return loikb_update_eq_constraint(S, c_id, Ai, bi, in_flags);  // hpp:250-253
  if (c_id < 1 || c_id >= S->ext_nj) { g_last_error = "constraint link id out of range"; return LOIKB_ERR_ARG; }
  if (S->nc_active >= S->nc) {
    g_last_error = "AddEqConstraint: no free constraint slot (loikb_options.eq_c_capacity)";
    return LOIKB_ERR_EQ_C_SIZE;
  }
  HIPCHK(hipSetDevice(S->device));
  ++S->inputs_epoch;
  int rc;
  const int k = S->nc_active;
  if ((rc = null_constraint_slots(S, k, k + 1))) return rc;  // the new constraint's dual starts at zero
  S->active_ids.push_back(c_id);
  ++S->nc_active;
  if ((rc = bind_constraint_slots(S))) return rc;
  if ((rc = update_eq_single(S, c_id, Ai, bi, in_flags))) {  // bis_inf_norm_ grows (hpp:281-283)
    S->active_ids.pop_back();
    --S->nc_active;
    (void)bind_constraint_slots(S);
    return rc;
  }
  return reset_home(S, RS_HCACHE);
}

int loikb_remove_eq_constraint(loikb_solver* S, int c_id)
{
  if (!S) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "RemoveEqConstraint() before SolveInit()"; return LOIKB_ERR_STATE; }
  int found = -1;
  for (int c = 0; c < S->nc_active; ++c)
    if (S->active_ids[c] == c_id) { found = c; break; }
  if (found < 0) {  // upstream prints this warning on stderr and returns (hpp:297-301)
    g_last_error = "WARNING RemoveEqConstraint: no constraint defined at link id, nothing to remove";
    return LOIKB_NOTHING_TO_REMOVE;
  }
  HIPCHK(hipSetDevice(S->device));
  ++S->inputs_epoch;
  int rc;
  // the entries behind it move down with their duals; the freed last slot becomes a null constraint
  if ((rc = edit_constraints(S, found, S->nc_active, 1))) return rc;
  for (int c = found; c + 1 < S->nc_active; ++c) {
    memcpy(S->A_host.data() + 36 * c, S->A_host.data() + 36 * (c + 1), 36 * sizeof(double));
    if (S->a_shared && (rc = upload_shared_A(S, S->A_host.data() + 36 * c, c))) return rc;
  }
  const double zero[36] = {0};
  memset(S->A_host.data() + 36 * (S->nc_active - 1), 0, 36 * sizeof(double));
  if ((rc = upload_shared_A(S, zero, S->nc_active - 1))) return rc;
  S->active_ids.erase(S->active_ids.begin() + found);
  --S->nc_active;
  if ((rc = bind_constraint_slots(S))) return rc;
  if ((rc = constraint_products(S, 0, S->nc, false))) return rc;  // bis_inf_norm_ recomputed over what is left (hpp:310-315)
  return reset_home(S, RS_HCACHE);
}

int loikb_num_eq_c(const loikb_solver* S) { return S ? S->nc_active : 0; }
int loikb_eq_c_capacity(const loikb_solver* S) { return S ? S->nc : 0; }
int loikb_active_constraint_ids(const loikb_solver* S, int* out, int cap)
{
  if (!S || (cap > 0 && !out)) return LOIKB_ERR_ARG;
  for (int c = 0; c < S->nc_active && c < cap; ++c) out[c] = S->active_ids[c];
  return S->nc_active;
}

int loikb_integrate(loikb_solver* S, double dt)
{
  if (!S) return LOIKB_ERR_ARG;
  if (!S->have_q) { g_last_error = "integrate: no configurations resident on the device yet"; return LOIKB_ERR_STATE; }
  ++S->inputs_epoch;
  HIPCHK(hipSetDevice(S->device));
  if (S->f32)
    hipLaunchKernelGGL(k_advance_q<float>, grid1(S->B), dim3(256), 0, S->stream, S->d_q, (const double*)nullptr, 0, S->nq,
                       S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles, dt);
  else
    hipLaunchKernelGGL(k_advance_q<double>, grid1(S->B), dim3(256), 0, S->stream, S->d_q, (const double*)nullptr, 0, S->nq,
                       S->d_jd, S->d_idx_q, S->L, S->B, S->home.tiles, dt);
  HIPCHK(hipGetLastError());
  return LOIKB_OK;
}

int loikb_synchronize(loikb_solver* S)
{
  if (!S) return LOIKB_ERR_ARG;
  HIPCHK(hipSetDevice(S->device));
  HIPCHK(hipDeviceSynchronize());
  return LOIKB_OK;
}

// ---- pass-level public methods of the reference (loik-loid-optimized.hpp:192-264) as a debug path, loik_passes.hpp
static PassParams pass_params(const loikb_solver_impl* S)
{
  PassParams P{};
  P.href_tab = S->d_href64;
  P.Hv_inf_norm = S->Hv_inf_norm;
  P.rho = S->opt.rho; P.mu0 = solve_mu0(S); P.mu_scale = S->opt.mu_equality_scale_factor;
  P.tol_abs = S->opt.tol_abs; P.tol_rel = S->opt.tol_rel; P.tol_primal_inf = S->opt.tol_primal_inf;
  P.tol_tail_solve = S->opt.tol_tail_solve;
  P.max_iter = S->opt.max_iter;
  P.mu_osqp = S->opt.mu_update_strat == LOIKB_MU_OSQP;
  P.a_shared = S->a_shared; P.bnd_shared = S->bnd_shared;
  return P;
}

// the pass state: allocated on first use, (re)loaded from the tiles whenever the solver state changed underneath it
static int ensure_pass_state(loikb_solver_impl* S, const PassParams& P)
{
  if (S->pass_active) return LOIKB_OK;
  const PassLayout PL = make_pass_layout(S->nj, S->nv, S->nc, S->B);
  if (!S->d_pass || PL.stride != S->PL.stride) {
    if (S->d_pass) HIPCHK(hipFree(S->d_pass));
    S->d_pass = nullptr;
    HIPCHK(hipMalloc((void**)&S->d_pass, sizeof(double) * (size_t)PL.stride * S->B));
    if (!S->d_pass_cslot) HIPCHK(hipMalloc((void**)&S->d_pass_cslot, sizeof(int) * S->nj));
  }
  S->PL = PL;
  std::vector<int> cs(S->nj, -1);
  for (int i = 1; i < S->nj; ++i) cs[i] = S->jd[i].cslot;
  HIPCHK(hipMemcpyAsync(S->d_pass_cslot, cs.data(), sizeof(int) * S->nj, hipMemcpyHostToDevice, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  if (S->f32)   // (the pass state is fp64 whatever the handle's precision)
    hipLaunchKernelGGL(k_pass_load<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, (const JointDesc*)S->d_jd,
                       (const float*)S->d_uni, S->PL, P, S->d_pass);
  else
    hipLaunchKernelGGL(k_pass_load<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, (const JointDesc*)S->d_jd,
                       (const double*)S->d_uni, S->PL, P, S->d_pass);
  HIPCHK(hipGetLastError());
  S->pass_active = true;
  return LOIKB_OK;
}

int loikb_pass(loikb_solver* S, int pass)
{
  if (!S || pass < PASS_BEGIN_ITERATION || pass > PASS_UPDATE_MU) return LOIKB_ERR_ARG;
  if (!S->have_problem) { g_last_error = "pass-level call before SolveInit()"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  const PassParams P = pass_params(S);
  int rc;
  if ((rc = ensure_pass_state(S, P))) return rc;
  hipLaunchKernelGGL(k_pass, grid1(S->B, 64), dim3(64), 0, S->stream, pass, S->PL, P, (const JointDesc*)S->d_jd,
                     (const int*)S->d_pass_cslot, S->d_pass);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

static int ensure_log(loikb_solver_impl* S);
static int finish_logged(loikb_solver_impl* S, const PassParams& P);
// a logged solve that the flat engine takes whole: fp64, a reference weight the engine of the robot's size takes (k_flat2 / k_flat1:
// any; k_flat: h I), one chunk, the batch goes to it directly
static bool logged_on_flat(const loikb_solver_impl* S)
{
  return S->opt.logging && !S->f32 && flat_applicable(S) && S->plan.nchunks == 1 && S->B >= 64 && S->B <= S->plan.tail_max &&
         !(S->opt.flags & (LOIKB_OPT_NO_H_CACHE | LOIKB_OPT_FIXED_ITERS));
}

// Solve with logging_ = true: SolverInfo lists filled.  On the flat engine when the solve qualifies (logged_on_flat), else the
// main loop on the plain pass implementation (k_pass_solve), whose results are then read from the pass state (loikb_get), like
// after loikb_pass.
// redo_reset: the reset_home flags that put the solver into the state this solve starts from, when that state can be put back (a cold
// Solve(), a solve that began with a cold data reset); 0 when it cannot (a warm start: the iterates the solve began with are gone).
// For the entries that begin with Reset(false) + FwdPassInit that is RS_DATA_COLD (w, z, nu, vis, fis, g) AND RS_Y (FwdPassInit's
// yis = Aty = 0, optimized.hxx:270-278) AND RS_HCACHE (fwd_pass_init's): the first attempt left its duals and factors behind.
static int run_logged(loikb_solver_impl* S, int redo_reset)
{
  int rc;
  if ((rc = ensure_log(S))) return rc;
  S->log_truncated = 0;
  bool on_flat = logged_on_flat(S);
  if (on_flat) {
    // the fast engine writes the lists itself (k_flat<.., LOG>).  An instance whose mu leaves the configured decades is finished by
    // k_tail, which writes no lists: its lists would end where it left the flat engine.  When that happens and the
    // solve's starting state can be put back, the whole solve is run again on the pass-by-pass implementation, which logs every
    // iteration of every instance; when it cannot (warm start), loikb_solver_info_truncated() says how many instances are short.
    HIPCHK(hipMemsetAsync(S->d_log_rows, 0, sizeof(int) * (size_t)S->B, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    if ((rc = run_main_loop(S))) return rc;
    S->have_log = true;
    if (S->stats.lean_escaped == 0) return LOIKB_OK;
    if (!redo_reset) { S->log_truncated = S->stats.lean_escaped; return LOIKB_OK; }
    if ((rc = reset_home(S, redo_reset))) return rc;
    if ((rc = ensure_log(S))) return rc;
    on_flat = false;
  }
  return run_pass_solve(S, true);
}

// the whole solve on the plain pass-by-pass implementation (one instance per thread, the data object of the reference field by field in
// HBM): a logged solve the flat engine does not take, and the engine of last resort of a robot k_solve cannot take (EnginePlan::solve_ok)
namespace {
int run_pass_solve(loikb_solver_impl* S, bool logged)
{
  int rc;
  if (S->opt.mu_update_strat != LOIKB_MU_DEFAULT && S->opt.mu_update_strat != LOIKB_MU_OSQP &&
      S->opt.mu_update_strat != LOIKB_MU_MAXEIGENVALUE && !(S->opt.flags & LOIKB_OPT_FIXED_ITERS)) {
    g_last_error = "[FirstOrderLoikOptimizedTpl::UpdateMu]: mu update strategy not supported";
    return LOIKB_ERR_MU_STRATEGY;
  }
  if ((rc = start_mu(S))) return rc;  // (run_main_loop does it on the other path)
  const PassParams P = pass_params(S);
  S->pass_active = false;  // the resets / updates of this solve went to the tiles: reload
  if ((rc = ensure_pass_state(S, P))) return rc;
  const int cap = logged ? S->log_rows_cap : 0;
  HIPCHK(hipEventRecord(S->ev_t0, S->stream));
  hipLaunchKernelGGL(k_pass_solve, grid1(S->B, 64), dim3(64), 0, S->stream, S->PL, P, (const JointDesc*)S->d_jd,
                     (const int*)S->d_pass_cslot, S->d_pass, logged ? S->d_log : (double*)nullptr, cap, logged ? S->d_log_rows : (int*)nullptr);
  HIPCHK(hipGetLastError());
  if ((rc = finish_logged(S, P))) return rc;
  if (!logged) S->have_log = false;
  return LOIKB_OK;
}
}  // namespace

// the SolverInfo lists of a logged solve: B x (max_iter - 1) rows x nine lists, zero beyond an instance's rows
static int ensure_log(loikb_solver_impl* S)
{
  const int cap = std::max(S->opt.max_iter - 1, 1);
  if (!S->d_log || cap != S->log_rows_cap) {
    if (S->d_log) HIPCHK(hipFree(S->d_log));
    S->d_log = nullptr;
    const size_t bytes = sizeof(double) * (size_t)S->B * cap * LOG_NLIST;
    if (hipMalloc((void**)&S->d_log, bytes) != hipSuccess) {
      (void)hipGetLastError();
      char buf[256];
      snprintf(buf, sizeof(buf), "logging: the SolverInfo lists need %.2f GB (batch %d x (max_iter - 1) %d x %d lists x 8 B)",
               bytes / 1e9, S->B, cap, (int)LOG_NLIST);
      g_last_error = buf;
      return LOIKB_ERR_HIP;
    }
    if (!S->d_log_rows) HIPCHK(hipMalloc((void**)&S->d_log_rows, sizeof(int) * S->B));
    S->log_rows_cap = cap;
  }
  HIPCHK(hipMemsetAsync(S->d_log, 0, sizeof(double) * (size_t)S->B * cap * LOG_NLIST, S->stream));
  return LOIKB_OK;
}

static int finish_logged(loikb_solver_impl* S, const PassParams& P)
{
  // the result goes back to the tiles too: a warm-started solve, loikb_integrate and the engines' getters continue from it
  if (S->f32) hipLaunchKernelGGL(k_pass_store<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, S->PL, P, (const double*)S->d_pass);
  else hipLaunchKernelGGL(k_pass_store<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, S->PL, P, (const double*)S->d_pass);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(S->ev_t1, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, S->ev_t0, S->ev_t1));
  S->stats = loikb_stats{};
  S->stats.launches = 1;
  S->stats.kernel_ms = ms; S->stats.total_ms = ms;
  {
    Chunk* C0 = &S->chunks[0];
    HIPCHK(hipMemsetAsync(C0->d_counters, 0, sizeof(unsigned int), S->stream));
    if (S->f32) hipLaunchKernelGGL(k_count_unfinished<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, S->B, C0->d_counters);
    else hipLaunchKernelGGL(k_count_unfinished<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, S->L, S->B, C0->d_counters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(C0->h_counters, C0->d_counters, sizeof(unsigned int), hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    S->stats.n_unfinished = (int)C0->h_counters[0];
  }
  S->have_log = true;
  return LOIKB_OK;
}

int loikb_solver_info_rows_cap(const loikb_solver* S) { return (S && S->have_log) ? S->log_rows_cap : 0; }
int loikb_solver_info_truncated(const loikb_solver* S) { return (S && S->have_log) ? S->log_truncated : 0; }

int loikb_get_solver_info(loikb_solver* S, int list, double* out, int out_rows_cap, int* rows_out)
{
  if (!S || list < 0 || list >= LOG_NLIST || !out || out_rows_cap < 1) return LOIKB_ERR_ARG;
  if (!S->have_log) { g_last_error = "no SolverInfo: create the solver with logging = 1 and solve"; return LOIKB_ERR_STATE; }
  HIPCHK(hipSetDevice(S->device));
  const int cap = S->log_rows_cap;  // rows per instance of the stored lists (max_iter - 1 AT THE TIME OF THE SOLVE)
  // (the lists are stored list-major and zero beyond rows[b]; the caller's rows may be shorter or longer than the stored ones)
  const int n = std::min(cap, out_rows_cap);
  if (out_rows_cap > cap) memset(out, 0, sizeof(double) * (size_t)S->B * out_rows_cap);
  HIPCHK(hipMemcpy2D(out, sizeof(double) * (size_t)out_rows_cap, S->d_log + (size_t)list * S->B * cap, sizeof(double) * (size_t)cap,
                     sizeof(double) * (size_t)n, (size_t)S->B, hipMemcpyDeviceToHost));
  if (rows_out) HIPCHK(hipMemcpy(rows_out, S->d_log_rows, sizeof(int) * (size_t)S->B, hipMemcpyDeviceToHost));
  return LOIKB_OK;
}

// loikb_get while the pass-level state is active: the members of the data object the passes maintain
static int pass_get(loikb_solver_impl* S, int field, void* out, bool to_dev)
{
  const PassLayout& L = S->PL;
  const int nb = S->nb;
  const int nl = S->ext_nj - 1;  // links of the caller's model (== nb unless it has multi-DoF joints)
  int off = -1, n = 0, skip = 0, scal = -1, width = 0;
  bool is_int = false;
  switch (field) {
  case LOIKB_F_Z: off = L.z; n = S->nv; break;
  case LOIKB_F_NU: off = L.nu; n = S->nv; break;
  case LOIKB_F_W: off = L.w; n = S->nv; break;
  case LOIKB_F_STF_PLUS_W: off = L.Stf; n = S->nv; break;
  case LOIKB_F_R: off = L.r; n = S->nv; break;
  case LOIKB_F_DINV: off = L.Dinv; n = nb; skip = 1; break;
  case LOIKB_F_VIS: off = L.vis; n = 6 * nl; width = 6; break;
  case LOIKB_F_FIS: off = L.fis; n = 6 * nl; width = 6; break;
  case LOIKB_F_G: off = L.g; n = 6 * nl; width = 6; break;
  case LOIKB_F_PIS: off = L.pis; n = 6 * nl; width = 6; break;
  case LOIKB_F_UDINV: off = L.UDinv; n = 6 * nb; skip = 6; break;   // per DoF (a column of the chain's elimination)
  case LOIKB_F_LIMI:
    if (nl != nb) return LOIKB_ERR_ARG;  // (M(q) of a multi-DoF joint is a product along its chain: k_limi on the tiles, same q)
    off = L.liMi; n = 12 * nb; skip = 12; break;
  case LOIKB_F_YIS: off = L.yis; n = 6 * S->nc_active; break;  // (the active constraints are the first slots)
  case LOIKB_F_ATY: off = L.Aty; n = 6 * S->nc_active; break;
  case LOIKB_F_HIS: n = 21 * nl; break;
  case LOIKB_F_ITER: scal = PS_ITER; is_int = true; break;
  case LOIKB_F_CONVERGED: scal = PS_CONVERGED; is_int = true; break;
  case LOIKB_F_PRIMAL_INFEASIBLE: scal = PS_PRIMAL_INF; is_int = true; break;
  case LOIKB_F_PRIMAL_RESIDUAL: scal = PS_PRIMAL; break;
  case LOIKB_F_DUAL_RESIDUAL: scal = PS_DUAL; break;
  case LOIKB_F_PRIMAL_RESIDUAL_TASK: scal = PS_PR_TASK; break;
  case LOIKB_F_PRIMAL_RESIDUAL_SLACK: scal = PS_PR_SLACK; break;
  case LOIKB_F_DUAL_RESIDUAL_V: scal = PS_DUAL_V; break;
  case LOIKB_F_DUAL_RESIDUAL_NU: scal = PS_DUAL_NU; break;
  case LOIKB_F_TOL_PRIMAL: scal = PS_TOL_P; break;
  case LOIKB_F_TOL_DUAL: scal = PS_TOL_D; break;
  case LOIKB_F_MU: scal = PS_MU; break;
  case LOIKB_F_MU_EQ: scal = PS_MU_EQ; break;
  case LOIKB_F_MU_INEQ: scal = PS_MU_IN; break;
  case LOIKB_F_DELTA_X_QP_INF_NORM: scal = PS_DX; break;
  case LOIKB_F_DELTA_Z_QP_INF_NORM: scal = PS_DZ_INF; break;
  case LOIKB_F_DELTA_Y_QP_INF_NORM: scal = PS_DYQP; break;
  case LOIKB_F_A_QP_T_DELTA_Y_QP_INF_NORM: scal = PS_ATDY; break;
  case LOIKB_F_UB_QP_T_DELTA_Y_QP_PLUS: scal = PS_UBP; break;
  case LOIKB_F_LB_QP_T_DELTA_Y_QP_MINUS: scal = PS_LBM; break;
  case LOIKB_F_DELTA_FIS_INF_NORM: scal = PS_DFIS_INF; break;
  case LOIKB_F_DELTA_YIS_INF_NORM: scal = PS_DYIS_INF; break;
  case LOIKB_F_DELTA_W_INF_NORM: scal = PS_DW_INF; break;
  case LOIKB_F_DELTA_VIS_INF_NORM: scal = PS_DVIS_INF; break;
  case LOIKB_F_DELTA_NU_INF_NORM: scal = PS_DNU_INF; break;
  case LOIKB_F_AV_INF_NORM: scal = PS_AV_INF; break;
  case LOIKB_F_NU_INF_NORM: scal = PS_NU_INF; break;
  case LOIKB_F_HREF_V_INF_NORM: scal = PS_HREFV_INF; break;
  case LOIKB_F_G_INF_NORM: scal = PS_G_INF; break;
  case LOIKB_F_STF_PLUS_W_INF_NORM: scal = PS_STF_INF; break;
  case LOIKB_F_PRIMAL_INFEASIBILITY_COND_1: scal = PS_C1; break;
  case LOIKB_F_PRIMAL_INFEASIBILITY_COND_2: scal = PS_C2; break;
  case LOIKB_F_TAIL_SOLVE_ITER: scal = PS_TAIL_IT; break;
  default:
    g_last_error = "this field is not part of the pass-level state";
    return LOIKB_ERR_ARG;
  }
  if (scal >= 0) { off = L.scal + scal; n = 1; }
  const size_t bytes = sizeof(double) * (size_t)S->B * n;
  int rc;
  if ((rc = ensure_stage(S, bytes))) return rc;
  double* dst = (to_dev && !is_int) ? (double*)out : (double*)S->d_stage;
  if (field == LOIKB_F_HIS)
    hipLaunchKernelGGL(k_pass_get_his, grid1(S->B), dim3(256), 0, S->stream, S->PL, (const double*)S->d_pass, (const int*)S->d_link_sel, nl, dst);
  else if (width)
    hipLaunchKernelGGL(k_pass_get_links, grid1(S->B), dim3(256), 0, S->stream, S->PL, (const double*)S->d_pass, off, width,
                       (const int*)S->d_link_sel, nl, dst);
  else hipLaunchKernelGGL(k_pass_get, grid1(S->B), dim3(256), 0, S->stream, S->PL, (const double*)S->d_pass, off, n, skip, dst);
  HIPCHK(hipGetLastError());
  if (is_int) {
    std::vector<double> tmp((size_t)S->B);
    HIPCHK(hipMemcpyAsync(tmp.data(), dst, bytes, hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    std::vector<int> iv((size_t)S->B);
    for (int b = 0; b < S->B; ++b) iv[b] = (int)tmp[b];
    if (to_dev) HIPCHK(hipMemcpy(out, iv.data(), sizeof(int) * S->B, hipMemcpyHostToDevice));
    else memcpy(out, iv.data(), sizeof(int) * S->B);
    return LOIKB_OK;
  }
  if (!to_dev) HIPCHK(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

int loikb_set_max_iter(loikb_solver* S, int v) { if (!S) return LOIKB_ERR_ARG; S->opt.max_iter = v; ++S->inputs_epoch; return LOIKB_OK; }
int loikb_set_rho(loikb_solver* S, double v)
{
  if (!S) return LOIKB_ERR_ARG;
  S->opt.rho = v;
  ++S->inputs_epoch;
  HIPCHK(hipSetDevice(S->device));
  return reset_home(S, RS_HCACHE);
}
int loikb_set_mu(loikb_solver* S, double v)
{
  if (!S) return LOIKB_ERR_ARG;
  S->opt.mu = v;
  ++S->inputs_epoch;
  S->seen_lo = 1 << 20; S->seen_hi = -(1 << 20); S->end_hist_n = 0;  // the decades are counted from mu0: the history no longer applies
  return LOIKB_OK;
}
int loikb_set_tol(loikb_solver* S, double a, double r)
{
  if (!S) return LOIKB_ERR_ARG;
  S->opt.tol_abs = a; S->opt.tol_rel = r;
  ++S->inputs_epoch;
  return LOIKB_OK;
}
int loikb_set_tol_primal_inf(loikb_solver* S, double v) { if (!S) return LOIKB_ERR_ARG; S->opt.tol_primal_inf = v; ++S->inputs_epoch; return LOIKB_OK; }
int loikb_set_tol_tail_solve(loikb_solver* S, double v) { if (!S) return LOIKB_ERR_ARG; S->opt.tol_tail_solve = v; ++S->inputs_epoch; return LOIKB_OK; }
int loikb_set_warm_start(loikb_solver* S, int v) { if (!S) return LOIKB_ERR_ARG; S->opt.warm_start = v; ++S->inputs_epoch; return LOIKB_OK; }

int loikb_batch(const loikb_solver* S) { return S ? S->B : 0; }
int loikb_nv(const loikb_solver* S) { return S ? S->nv : 0; }
int loikb_njoints(const loikb_solver* S) { return S ? S->ext_nj : 0; }

const char* loikb_plan_string(loikb_solver* S)
{
  static thread_local std::string out;
  if (!S) return "";
  char buf[512];
  const EnginePlan& pl = S->plan;
  if (pl.flat && (flat_applicable(S) || !S->have_problem)) {
    const bool split = S->tune.flat_split && S->flat.G == F2G && S->flat.nanc <= FLAT_NA_SMALL && !S->f32;
    const bool one = S->tune.flat_split && S->flat.G == WAVE && !S->f32;
    snprintf(buf, sizeof(buf), "k_fslots + %s (no loops over the tree levels%s) for whole batches up to %d instances (%d wavefronts per "
             "CU, decades mu0*10^%d..%d, %d ancestors per joint, %d scan steps, %d jump rounds)%s; k_solve above that; %d chunk(s)",
             split ? "k_flat2" : one ? "k_flat1" : "k_flat",
             split ? "; two lanes per joint, one instance per wavefront" : one ? "; one instance per wavefront" : "", pl.tail_max,
             split ? std::min(4 * S->tune.flat_split_wpe, pl.flat_waves_cu * 2) : one ? std::min(8, pl.flat_waves_cu) : pl.flat_waves_cu,
             pl.kexp_lo, pl.kexp_lo + pl.ndec - 1,
             S->flat.nanc, S->flat.nscan, S->flat.njmp,
             (split || one) ? ", any reference cost" : S->have_problem ? "" : " when H_ref = h I", pl.nchunks);
  }
  else if (pl.lean)
    snprintf(buf, sizeof(buf), "k_hslots + k_lean for whole batches up to %d instances (%d wavefronts per CU in workgroups of %d, "
             "decades mu0*10^%d..%d, time slice %d); k_solve above that; %d chunk(s)", pl.tail_max, pl.lean_waves_cu,
             pl.lean_wg_waves, pl.kexp_lo, pl.kexp_lo + pl.ndec - 1, S->tune.lean_slice, pl.nchunks);
  else
    snprintf(buf, sizeof(buf), "k_solve (team of %d), hand-over to k_tail at %d live instances; %d chunk(s); no k_lean: %s",
             S->sched[1].nw, pl.tail_max, pl.nchunks, pl.why_not_lean);
  out = buf;
  if (pl.flat && flat_any_mu(S) && (flat_applicable(S) || !S->have_problem))
    out += "; OSQP penalty rule: mu is off the decade grid -- k_fslots builds mu0's slot only, the iteration kernel builds W / Dinv in-wave at every change of mu";
  else if (pl.flat && S->tune.flat_build && (flat_applicable(S) || !S->have_problem))
    out += S->tune.flat_build == 2 ? "; lazily populated decade table (LOIKB_FLAT_BUILD=2, the default): on time-sliced launches k_fslots builds the decades 97 % of the "
                                     "previous solve's instances ended within (a handle's first solve of 49 152+ instances: the five from mu0's upwards), an instance "
                                     "that goes further builds its slot in-wave, once"
                                   : "; LOIKB_FLAT_BUILD=1: a decade of the table k_fslots did not build (LOIKB_FLAT_WINDOW=lo,n), or beyond the table, is built "
                                     "in-wave by the instance that gets there (no hand-over to k_tail)";
  if (!pl.flat && *pl.why_not_flat) out += std::string("; no k_flat: ") + pl.why_not_flat;
  else if (pl.flat && S->have_problem && !flat_applicable(S))
    out += pl.lean ? "; no flat engine for this problem (per-link reference weights): k_hslots + k_lean take its place"
                   : "; no flat engine for this problem: per-link reference weights";
  if ((pl.lean || pl.flat) && S->tune.lean_adapt && S->seen_hi >= S->seen_lo) {
    char b2[160];
    snprintf(b2, sizeof(b2), "; decades visited by this handle's solves so far: %d..%d (the next solve builds those +-1)", S->seen_lo, S->seen_hi);
    out += b2;
  }
  if (pl.flat && (flat_applicable(S) || !S->have_problem) && flat_takes_diagonal(S)) {
    const int q = flat_slice_for(S, S->B, false);
    if (q > 0) {
      char b3[200];
      snprintf(b3, sizeof(b3), "; launches without an order (a handle's first solve of its inputs) are time-sliced: %d iterations, then %d, "
               "while other instances wait", q & 0xffff, ((q >> 16) & 0x3fff) ? ((q >> 16) & 0x3fff) : (q & 0xffff));
      out += b3;
    }
  }
  {   // device memory the handle holds besides the instances' tiles (the kept getter scratch was nowhere to be seen)
    size_t fs = 0, pk = 0, hs = 0;
    for (const Chunk& C : S->chunks) { fs += C.fslots_bytes; pk += C.park_bytes; hs += C.hslots_bytes; }
    const size_t gs = S->getscr_bytes[0] + S->getscr_bytes[1];
    if (fs + pk + hs + gs > 0) {
      char b4[240];
      snprintf(b4, sizeof(b4), "; device buffers: decade slots %.0f MB, park records %.0f MB, getters' scratch %.0f MB (kept between calls; "
               "given back when the slots or park records need the room)", (fs + hs) / 1048576.0, pk / 1048576.0, gs / 1048576.0);
      out += b4;
    }
  }
  if (!pl.solve_ok) {
    char b5[400];
    snprintf(b5, sizeof(b5), "a tree too bushy for k_solve (its leaf->root hand-over slots would need %.0f KB of a CU's 160 KB of LDS): %s; ",
             pl.solve_lds_need / 1024.0, bushy_goes_on_chip(S) ? "whole batches go to the on-chip engines, which have no such slots"
                                                               : "every solve runs on the plain pass-by-pass implementation (k_pass_solve), the engine of last resort");
    out = b5 + out;
  }
  if (S->opt.logging && logged_on_flat(S)) out = "logging = 1: the flat engine writes the SolverInfo lists; " + out;
  else if (S->opt.logging) out = "logging = 1: every solve runs on the plain pass-by-pass implementation (k_pass_solve) and fills SolverInfo; without it: " + out;
  if (S->per_link && flat_applicable(S)) out += "; per-link references in force (UpdateReferences): the flat engine reads the links' table";
  else if (pl.lean && S->per_link) out += "; per-link references in force (UpdateReferences): k_hslots + k_lean in their per-link instantiations until the next SolveInit";
  return out.c_str();
}

int loikb_get_stats(loikb_solver* S, loikb_stats* out)
{
  if (!S || !out) return LOIKB_ERR_ARG;
  *out = S->stats;
  return LOIKB_OK;
}

// rows of the members a solve leaves for the caller (the same maps as loikb_get's: ONE definition)
static bool result_rows(const loikb_solver_impl* S, int field, std::vector<int>& rm)
{
  const Layout& L = S->L;
  const int nb = S->nb, nl = S->ext_nj - 1;
  auto per_joint = [&](int pair, int half) { for (int j = 0; j < nb; ++j) rm.push_back((j * JREC + pair) * 2 + half); };
  auto per_joint_vec = [&](int pair, int n) {  // body i of the caller's model lives in device joint link_of[i]
    for (int i = 1; i <= nl; ++i)
      for (int k = 0; k < n; ++k) rm.push_back(((S->link_of[i] - 1) * JREC + pair + k / 2) * 2 + (k & 1));
  };
  switch (field) {
  case LOIKB_F_Z: per_joint(JP_WZ, 1); return true;
  case LOIKB_F_NU: per_joint(JP_NUS, 0); return true;
  case LOIKB_F_W: per_joint(JP_WZ, 0); return true;
  case LOIKB_F_VIS: per_joint_vec(JP_V, 6); return true;
  case LOIKB_F_FIS: per_joint_vec(JP_F, 6); return true;
  case LOIKB_F_YIS:
    for (int c = 0; c < S->nc_active; ++c)  // (the active constraints are the first slots; the null ones behind them hold zeros)
      for (int k = 0; k < 6; ++k) rm.push_back((L.off_c + c * L.crec + CP_Y + k / 2) * 2 + (k & 1));
    return true;
  case -1:   // (loikb_get_results' scalar block: LOIKB_RES_NSCALARS rows, the maps loikb_get uses for the single fields)
    for (int f = LOIKB_F_PRIMAL_RESIDUAL; f <= LOIKB_F_TAIL_SOLVE_ITER; ++f) {
      const int idx = f - LOIKB_F_PRIMAL_RESIDUAL;
      rm.push_back(f == LOIKB_F_MU ? (L.off_s + SP_MU) * 2 : (L.off_s + SP_SCAL + idx / 2) * 2 + (idx & 1));
    }
    rm.push_back((L.off_s + SP_BI) * 2 + 1);   // iter
    rm.push_back((L.off_s + SP_ST) * 2);       // status bits
    rm.push_back((L.off_s + SP_FLIP) * 2);     // mu updates
    return true;
  default: return false;
  }
}

// z, nu, w, vis, fis, yis of the whole batch in ONE call: one gather launch into pinned host memory and one synchronisation for a batch whose
// results fit LOIKB_RESULTS_FUSED_BYTES (default 4 MiB); larger batches, and a handle in the middle of pass-level calls, go field by field
// through loikb_get (whose copies are DMA transfers of their own).
int loikb_get_results(loikb_solver* S, unsigned int mask, double* z, double* nu, double* w, double* vis, double* fis, double* yis, double* scalars)
{
  if (!S) return LOIKB_ERR_ARG;
  constexpr int NF = 7;
  static_assert(LOIKB_RES_NSCALARS == NSCAL + 3 && LOIKB_RES_SCALAR_ITER == NSCAL, "scalar block out of step with the scalar record");
  static const int fields[NF] = {LOIKB_F_Z, LOIKB_F_NU, LOIKB_F_W, LOIKB_F_VIS, LOIKB_F_FIS, LOIKB_F_YIS, -1};
  double* outs[NF] = {z, nu, w, vis, fis, yis, scalars};
  if (mask & ~127u) { g_last_error = "loikb_get_results: unknown bits in the mask"; return LOIKB_ERR_ARG; }
  for (int f = 0; f < NF; ++f)
    if ((mask & (1u << f)) && !outs[f]) { g_last_error = "loikb_get_results: a requested member has no destination"; return LOIKB_ERR_ARG; }
  if (!mask) return LOIKB_OK;
  HIPCHK(hipSetDevice(S->device));
  static const size_t fused_max = [] { const char* e = getenv("LOIKB_RESULTS_FUSED_BYTES"); return e ? (size_t)atoll(e) : (size_t)4 << 20; }();
  const int key[6] = {(int)mask, S->nc_active, S->nb, S->ext_nj - 1, S->L.off_c, S->L.crec};
  int rc;
  loikb_solver_impl::ResMap* R = nullptr;
  for (auto& rmap : S->resmaps) if (rmap.d && memcmp(key, rmap.key, sizeof(key)) == 0) R = &rmap;
  if (!R) {
    R = &S->resmaps[S->resmap_next];
    S->resmap_next = (S->resmap_next + 1) % 4;
    std::vector<int> rm;
    int off = 0;
    for (int f = 0; f < NF; ++f) {
      R->off[f] = off;
      if (mask & (1u << f)) result_rows(S, fields[f], rm);
      off = (int)rm.size();
    }
    R->off[NF] = off;
    R->n = off;
    if (R->d) { HIPCHK(hipStreamSynchronize(S->stream)); HIPCHK(hipFree(R->d)); R->d = nullptr; }
    R->key[0] = -1;
    HIPCHK(hipMalloc((void**)&R->d, sizeof(int) * (size_t)std::max(off, 1)));
    HIPCHK(hipMemcpy(R->d, rm.data(), sizeof(int) * (size_t)off, hipMemcpyHostToDevice));   // (synchronous: rm is a local)
    memcpy(R->key, key, sizeof(key));
  }
  const int n = R->n;
  const size_t bytes = sizeof(double) * (size_t)S->B * (size_t)n;
  if (S->pass_active || bytes > fused_max || n == 0) {
    for (int f = 0; f < NF; ++f) {
      if (!(mask & (1u << f))) continue;
      if (fields[f] == LOIKB_F_YIS && S->nc_active == 0) continue;
      if (fields[f] == -1) {   // (the scalar block, field by field: a batch too large for the fused gather should ask for the scalars it needs instead)
        std::vector<double> col((size_t)S->B);
        std::vector<int> icol((size_t)S->B);
        for (int k = 0; k < LOIKB_RES_NSCALARS; ++k) {
          if (k < NSCAL) { if ((rc = loikb_get(S, LOIKB_F_PRIMAL_RESIDUAL + k, col.data(), 0))) return rc; }
          else {
            if ((rc = loikb_get(S, k == LOIKB_RES_SCALAR_ITER ? LOIKB_F_ITER : k == LOIKB_RES_SCALAR_STATUS ? LOIKB_F_STATUS : LOIKB_F_MU_UPDATES, icol.data(), 0))) return rc;
            if (k == LOIKB_RES_SCALAR_STATUS) {
              // (in the middle of pass-level calls the flags live in the pass state, the raw word in the tiles: bits 1 and 2 from the flags' own getters)
              std::vector<int> fl((size_t)S->B);
              for (int b = 0; b < S->B; ++b) icol[(size_t)b] &= ~(ST_CONVERGED | ST_PRIMAL_INF);
              if ((rc = loikb_get(S, LOIKB_F_CONVERGED, fl.data(), 0))) return rc;
              for (int b = 0; b < S->B; ++b) icol[(size_t)b] |= fl[(size_t)b] ? ST_CONVERGED : 0;
              if ((rc = loikb_get(S, LOIKB_F_PRIMAL_INFEASIBLE, fl.data(), 0))) return rc;
              for (int b = 0; b < S->B; ++b) icol[(size_t)b] |= fl[(size_t)b] ? ST_PRIMAL_INF : 0;
            }
            for (int b = 0; b < S->B; ++b) col[(size_t)b] = (double)icol[(size_t)b];
          }
          for (int b = 0; b < S->B; ++b) scalars[(size_t)b * LOIKB_RES_NSCALARS + k] = col[(size_t)b];
        }
        continue;
      }
      if ((rc = loikb_get(S, fields[f], outs[f], 0))) return rc;
    }
    return LOIKB_OK;
  }
  if (bytes > S->h_res_bytes) {
    if (S->h_res) { HIPCHK(hipStreamSynchronize(S->stream)); HIPCHK(hipHostFree(S->h_res)); S->h_res = nullptr; S->h_res_bytes = 0; }
    HIPCHK(hipHostMalloc((void**)&S->h_res, bytes));
    S->h_res_bytes = bytes;
  }
  // one element per thread (a thread per instance walks its ~400 rows one dependent load after the other: 0.25 ms whatever the batch);
  // a handful of instances straight into the pinned buffer, more through the staging buffer and ONE copy
  const bool direct = bytes <= ((size_t)256 << 10);
  double* dst = S->h_res;
  if (!direct) {
    if ((rc = ensure_stage(S, bytes))) return rc;
    dst = (double*)S->d_stage;
  }
  {
    const long long total = (long long)S->B * n;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (S->f32) hipLaunchKernelGGL(k_download_elems<float>, grid, dim3(256), 0, S->stream, S->home.tiles, S->L, (const int*)R->d, n, S->B, dst);
    else hipLaunchKernelGGL(k_download_elems<double>, grid, dim3(256), 0, S->stream, S->home.tiles, S->L, (const int*)R->d, n, S->B, dst);
    HIPCHK(hipGetLastError());
  }
  if (!direct) HIPCHK(hipMemcpyAsync(S->h_res, S->d_stage, bytes, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  for (int f = 0; f < NF; ++f) {
    if (!(mask & (1u << f))) continue;
    const int o = R->off[f], nf = R->off[f + 1] - o;
    if (nf == 0) continue;
    for (int b = 0; b < S->B; ++b) memcpy(outs[f] + (size_t)b * nf, S->h_res + (size_t)b * n + o, sizeof(double) * (size_t)nf);
    if (fields[f] == -1)   // (the status word carries engine-internal bits above the four the API names: LOIKB_F_STATUS's mask)
      for (int b = 0; b < S->B; ++b) { double& x = outs[f][(size_t)b * nf + LOIKB_RES_SCALAR_STATUS]; x = (double)((int)x & (ST_CONVERGED | ST_PRIMAL_INF | ST_TAIL | ST_DONE)); }
  }
  return LOIKB_OK;
}

int loikb_get(loikb_solver* S, int field, void* out, int out_flags)
{
  if (!S || !out) return LOIKB_ERR_ARG;
  HIPCHK(hipSetDevice(S->device));
  const bool to_dev = out_flags & LOIKB_OUT_DEVICE;
  if ((field == LOIKB_F_HIS || field == LOIKB_F_PRIMAL_RESIDUAL_VEC || field == LOIKB_F_DUAL_RESIDUAL_VEC || field == LOIKB_F_UDINV ||
       field == LOIKB_F_PIS) && !S->have_problem) {
    g_last_error = "this member is built from the problem's reference cost: call SolveInit() first";
    return LOIKB_ERR_STATE;
  }
  if (S->pass_active && field != LOIKB_F_Q) {
    const int rc_pass = pass_get(S, field, out, to_dev);
    if (rc_pass != LOIKB_ERR_ARG) return rc_pass;
    // (not a member the pass state keeps -- status bits, mu updates, residual vectors: served from the tiles below, which a
    //  logged solve leaves in sync, k_pass_store)
  }
  if (field == LOIKB_F_Q) {
    if (!S->have_q) { g_last_error = "no configurations resident on the device yet"; return LOIKB_ERR_STATE; }
    HIPCHK(hipMemcpyAsync(out, S->d_q, sizeof(double) * (size_t)S->B * S->nq,
                          to_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    return LOIKB_OK;
  }
  const Layout& L = S->L;
  const int nb = S->nb;         // device joints == model.nv: the per-DoF fields
  const int nl = S->ext_nj - 1; // bodies of the caller's model: the per-link fields
  std::vector<int> rm;
  bool is_int = false;
  int mask = 0;
  auto per_joint = [&](int pair, int half) { for (int j = 0; j < nb; ++j) rm.push_back((j * JREC + pair) * 2 + half); };
  auto per_joint_vec = [&](int pair, int n) {  // body i of the caller's model lives in device joint link_of[i]
    for (int i = 1; i <= nl; ++i)
      for (int k = 0; k < n; ++k) rm.push_back(((S->link_of[i] - 1) * JREC + pair + k / 2) * 2 + (k & 1));
  };
  auto per_constraint_vec = [&](int pair) {
    for (int c = 0; c < S->nc_active; ++c)  // (the active constraints are the first slots; the null ones behind them hold zeros)
      for (int k = 0; k < 6; ++k) rm.push_back((L.off_c + c * L.crec + pair + k / 2) * 2 + (k & 1));
  };
  switch (field) {
  case LOIKB_F_Z: case LOIKB_F_NU: case LOIKB_F_W: case LOIKB_F_VIS: case LOIKB_F_FIS: case LOIKB_F_YIS: result_rows(S, field, rm); break;
  case LOIKB_F_STF_PLUS_W: per_joint(JP_NUS, 1); break;
  case LOIKB_F_R: per_joint(JP_R, 0); break;
  case LOIKB_F_DINV: per_joint(JP_R, 1); break;
  case LOIKB_F_G: per_joint_vec(JP_G, 6); break;
  case LOIKB_F_PIS: break;  // the hot path keeps p_i^base: the accumulated p_i is rebuilt below
  case LOIKB_F_UDINV:  // per DoF (== per joint for 1-DoF joints; one column of the chain's elimination otherwise)
    for (int j = 0; j < nb; ++j)
      for (int k = 0; k < 6; ++k) rm.push_back((j * JREC + JP_UD + k / 2) * 2 + (k & 1));
    break;
  case LOIKB_F_HIS: break;  // not materialised by the hot path: rebuilt below

  case LOIKB_F_ATY: per_constraint_vec(CP_ATY); break;
  case LOIKB_F_LIMI: break;
  case LOIKB_F_PRIMAL_RESIDUAL_VEC: case LOIKB_F_DUAL_RESIDUAL_VEC: break;  // rebuilt below
  case LOIKB_F_ITER: is_int = true; rm.push_back((L.off_s + SP_BI) * 2 + 1); break;
  case LOIKB_F_STATUS: is_int = true; mask = -(ST_CONVERGED | ST_PRIMAL_INF | ST_TAIL | ST_DONE); rm.push_back((L.off_s + SP_ST) * 2); break;
  case LOIKB_F_CONVERGED: is_int = true; mask = ST_CONVERGED; rm.push_back((L.off_s + SP_ST) * 2); break;
  case LOIKB_F_PRIMAL_INFEASIBLE: is_int = true; mask = ST_PRIMAL_INF; rm.push_back((L.off_s + SP_ST) * 2); break;
  case LOIKB_F_MU: rm.push_back((L.off_s + SP_MU) * 2); break;  // per-instance mu_ (== mu0 right after ResetSolver)
  case LOIKB_F_MU_UPDATES: is_int = true; rm.push_back((L.off_s + SP_FLIP) * 2); break;
  default:
    static_assert(LOIKB_F_TAIL_SOLVE_ITER - LOIKB_F_PRIMAL_RESIDUAL + 1 == NSCAL, "scalar field ids out of sync");
    if (field >= LOIKB_F_PRIMAL_RESIDUAL && field <= LOIKB_F_TAIL_SOLVE_ITER) {
      const int idx = field - LOIKB_F_PRIMAL_RESIDUAL;  // same order as the SC_* enum
      rm.push_back((L.off_s + SP_SCAL + idx / 2) * 2 + (idx & 1));
    } else {
      return LOIKB_ERR_ARG;
    }
  }
  const bool resvec = field == LOIKB_F_PRIMAL_RESIDUAL_VEC || field == LOIKB_F_DUAL_RESIDUAL_VEC;
  const int n = field == LOIKB_F_LIMI ? 12 * nl : field == LOIKB_F_HIS ? 21 * nl : field == LOIKB_F_PIS ? 6 * nl
                : resvec ? 6 * nl + nb : (int)rm.size();
  // the rebuild kernels work on the device tree; with multi-DoF joints the caller's bodies are selected afterwards
  const bool select = nl != nb && (field == LOIKB_F_HIS || field == LOIKB_F_PIS);
  double* final_dst = nullptr;
  void* d_tmp = nullptr;
  const size_t bytes = (is_int ? sizeof(int) : sizeof(double)) * (size_t)S->B * n;
  double* dst = (double*)out;
  int rc;
  if (!to_dev) {
    if ((rc = ensure_stage(S, bytes))) return rc;
    dst = (double*)S->d_stage;
  }
  if (select) {
    final_dst = dst;
    if ((rc = ensure_getscr(S, 0, sizeof(double) * (size_t)S->B * nb * (field == LOIKB_F_HIS ? 21 : 6)))) return rc;
    d_tmp = S->d_getscr[0];
    dst = (double*)d_tmp;
  }
  if ((field == LOIKB_F_UDINV || field == LOIKB_F_PIS) && S->ud_stale) {
    // instances left by the flat engine carry neither UDinv nor pis (tag -2 in their scalar record): rebuilt in place
    if ((rc = ensure_getscr(S, 1, S->esz * (size_t)S->B * nb * 21))) return rc;
    void* d_scr = S->d_getscr[1];
    if (S->f32)
      hipLaunchKernelGGL(k_rebuild_ud<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, (const JointDesc*)S->d_jd,
                         (const float*)S->d_uni, (float)S->opt.rho, (float)S->opt.mu_equality_scale_factor, (const float*)S->d_href,
                         (int)S->a_shared, S->B, (float*)d_scr);
    else
      hipLaunchKernelGGL(k_rebuild_ud<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, (const JointDesc*)S->d_jd,
                         (const double*)S->d_uni, (double)S->opt.rho, (double)S->opt.mu_equality_scale_factor,
                         (const double*)S->d_href, (int)S->a_shared, S->B, (double*)d_scr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(S->stream));
    S->ud_stale = false;
  }
  if (field == LOIKB_F_HIS) {
    if (S->f32)
      hipLaunchKernelGGL(k_rebuild_his<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd,
                         (const float*)S->d_uni, (float)S->opt.rho, (float)S->opt.mu_equality_scale_factor,
                         (const float*)S->d_href, (int)S->a_shared, S->B, dst);
    else
      hipLaunchKernelGGL(k_rebuild_his<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd,
                         (const double*)S->d_uni, (double)S->opt.rho, (double)S->opt.mu_equality_scale_factor,
                         (const double*)S->d_href, (int)S->a_shared, S->B, dst);
  } else if (field == LOIKB_F_PIS) {
    if (S->f32) hipLaunchKernelGGL(k_rebuild_pis<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd, S->B, dst);
    else hipLaunchKernelGGL(k_rebuild_pis<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd, S->B, dst);
  } else if (resvec) {
    const int dual = field == LOIKB_F_DUAL_RESIDUAL_VEC;
    if (S->f32) hipLaunchKernelGGL(k_residual_vecs<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd, (const float*)S->d_uni, S->d_href64, (int)S->a_shared, S->d_link_sel, nl, S->B, dual, dst);
    else hipLaunchKernelGGL(k_residual_vecs<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd, (const double*)S->d_uni, S->d_href64, (int)S->a_shared, S->d_link_sel, nl, S->B, dual, dst);
  } else if (field == LOIKB_F_LIMI) {
    if (S->f32) hipLaunchKernelGGL(k_limi<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd, S->d_first_sel, S->d_link_sel, nl, S->B, dst);
    else hipLaunchKernelGGL(k_limi<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_jd, S->d_first_sel, S->d_link_sel, nl, S->B, dst);
  } else {
    if ((rc = set_rowmap(S, rm))) return rc;
    const size_t dbytes = sizeof(double) * (size_t)S->B * (size_t)n;
    if (!to_dev && n > 0 && dbytes <= ((size_t)256 << 10)) {
      // a small batch, a host destination: one element per thread straight into the pinned buffer (loikb_get_results' path) and ONE synchronisation --
      // no copy operation; the int fields' masks (k_download_rows') applied here
      if (dbytes > S->h_res_bytes) {
        if (S->h_res) { HIPCHK(hipStreamSynchronize(S->stream)); HIPCHK(hipHostFree(S->h_res)); S->h_res = nullptr; S->h_res_bytes = 0; }
        HIPCHK(hipHostMalloc((void**)&S->h_res, (size_t)256 << 10));
        S->h_res_bytes = (size_t)256 << 10;
      }
      const long long total = (long long)S->B * n;
      const dim3 g((unsigned)((total + 255) / 256));
      if (S->f32) hipLaunchKernelGGL(k_download_elems<float>, g, dim3(256), 0, S->stream, S->home.tiles, L, (const int*)S->d_rowmap, n, S->B, S->h_res);
      else hipLaunchKernelGGL(k_download_elems<double>, g, dim3(256), 0, S->stream, S->home.tiles, L, (const int*)S->d_rowmap, n, S->B, S->h_res);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(S->stream));
      if (is_int) {
        int* o = (int*)out;
        for (long long i = 0; i < total; ++i) {
          const int v = (int)S->h_res[i];
          o[i] = mask > 0 ? ((v & mask) ? 1 : 0) : (mask < 0 ? (v & -mask) : v);
        }
      } else {
        memcpy(out, S->h_res, dbytes);
      }
      return LOIKB_OK;
    }
    if (S->f32)
      hipLaunchKernelGGL(k_download_rows<float>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_rowmap, n,
                         S->B, dst, (int)is_int, mask);
    else
      hipLaunchKernelGGL(k_download_rows<double>, grid1(S->B), dim3(256), 0, S->stream, S->home.tiles, L, S->d_rowmap, n,
                         S->B, dst, (int)is_int, mask);
  }
  HIPCHK(hipGetLastError());
  if (select) {
    hipLaunchKernelGGL(k_select_rows, grid1(S->B), dim3(256), 0, S->stream, (const double*)d_tmp, nb,
                       field == LOIKB_F_HIS ? 21 : 6, S->d_link_sel, nl, S->B, final_dst);
    HIPCHK(hipGetLastError());
    dst = final_dst;
  }
  if (!to_dev) HIPCHK(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return LOIKB_OK;
}

#ifdef LOIKB_TAIL_PROF
// diagnostic build only: cycles per phase of wavefront 0 of the last tail launch + its iteration count
int loikb_debug_wave_dbg(unsigned long long* out)
{
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(loikb::g_wave_dbg), sizeof(unsigned long long) * 4096 * 6));
  return LOIKB_OK;
}
int loikb_debug_tail_prof(unsigned long long* out)
{
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(loikb::g_tail_prof), sizeof(unsigned long long) * 32));
#ifdef LOIKB_FLAT_SEPARATE_TU
  // (two units: the flat kernels wrote THEIR copy -- taken when it holds a launch; the callers profile one engine at a time)
  unsigned long long f[32];
  if (loikb_flat_prof_read(f, 0, 0)) return LOIKB_ERR_HIP;
  if (f[8]) memcpy(out, f, sizeof(f));
#endif
  return LOIKB_OK;
}
int loikb_debug_tail_prof_all(unsigned long long* out, int reset)   // the phases summed over all wavefronts since the last reset
{
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(loikb::g_tail_prof_all), sizeof(unsigned long long) * 32));
  if (reset) {
    unsigned long long z[32] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(loikb::g_tail_prof_all), z, sizeof(z)));
  }
#ifdef LOIKB_FLAT_SEPARATE_TU
  unsigned long long f[32];
  if (loikb_flat_prof_read(f, 1, reset)) return LOIKB_ERR_HIP;
  for (int k = 0; k < 32; ++k) out[k] += f[k];   // (sums since the last reset: one of the two copies is zero unless both kinds of kernel ran)
#endif
  return LOIKB_OK;
}
#endif

}  // extern "C"
