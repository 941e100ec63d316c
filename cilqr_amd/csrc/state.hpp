// Device-side state of the batched CILQR solver (gfx950).
//
// HBM layout: every per-problem tensor is batch-fastest ("slot" = position of a problem in
// the arena, stride Bcap) so that a wavefront whose 64 lanes hold 64 consecutive problems
// issues fully coalesced loads.  fp64 values are stored in pairs (double2 per slot) wherever
// the row count is even, so a lane moves 16 B per load and a wave 1 KiB per instruction:
//
//   X      [2][K][3][Bcap] double2   (x,y) (theta,v) (a,delta); two buffers, `cur[slot]` selects
//   U      [2][N][1][Bcap] double2   (jerk, delta_rate)
//   goals  [K][3][Bcap]    double2
//   cor    [K][cmax][3][Bcap] double shrunk+normalised planes;  ccnt [K][Bcap] int
//   lin    [N][17][Bcap]   double2   per step, only the entries that are not structurally constant:
//                                      A(0,2..5) A(1,2..5) A(2,3..5) B(2,1) | lx 6 | lu 2 |
//                                      lxx(0..2,0..2) lxx(3,3) lxx(4,4) lxx(5,5) | luu(0,0) luu(1,1)
//                                    (A = I + strictly-upper terms, A(3,4) = dt; B(3,0) = dt^2/2,
//                                     B(4,0) = B(5,1) = dt; all other entries are exact zeros in the
//                                     reference too: vehicle_model.cc:61-85, ilqr_optimizer.cc:642-650)
//   term   [9][Bcap]       double2   lx_N 6 | lxx_N (3x3 block + 3 diagonal)
//   gains  [N][7][Bcap]    double2   K (2x6 row-major) | k (2)
//   part   [K][3][Bcap]    double2   per-knot cost partials (J, bounds) of the state | of the control | (corridor, lane)
//   hist   [max_iter+1][5][Bcap] double
//
// Lane tables are shared by the batch and read through wave-uniform (scalar) loads.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cilqr {

constexpr int kNX = 6;
constexpr int kNU = 2;
constexpr int kLinPairs = 17;   // 34 stored doubles per step (of 96 dense)
constexpr int kTermPairs = 9;   // 18 stored doubles (of 42 dense)
constexpr int kGainPairs = 7;   // 14 doubles
constexpr int kPartPairs = 3;
constexpr int kNumAlpha = 11;
constexpr int kMaxDiscs = 16;
constexpr int kLaneFields = 11;  // a b c | sx sy | ux uy | len | ex ey | pad (an odd row stride: rows of different segments start in different LDS banks)
constexpr int kGridCellBytes = 16;
#ifndef CILQR_GRID_CELLS_LOG2
#define CILQR_GRID_CELLS_LOG2 16
#endif
constexpr int kGridMaxCells = 1 << CILQR_GRID_CELLS_LOG2;  // per side
constexpr int kGridFullScan = 255;      // cell marker: too many candidates, scan every segment

// pair rows of a step inside `lin`
//  0 (a02,a03) 1 (a04,a05) 2 (a12,a13) 3 (a14,a15) 4 (a23,a24) 5 (a25,b21)
//  6 (lx0,lx1) 7 (lx2,lx3) 8 (lx4,lx5) 9 (lu0,lu1)
// 10 (h00,h01) 11 (h02,h10) 12 (h11,h12) 13 (h20,h21) 14 (h22,h33) 15 (h44,h55) 16 (luu00,luu11)
// pair rows of `term`: 0..2 lx_N, 3..8 = rows 10..15 above
constexpr int kRowA = 0, kRowLx = 6, kRowLu = 9, kRowH = 10, kRowLuu = 16;
// dense byte counts the roofline is quoted against: 96 read + 14 written doubles
constexpr int kDenseDoublesPerStep = 110;
constexpr int kDenseDoublesTerminal = 44;

struct Params {
  int N, K;
  int num_of_disc;
  int max_iter;
  double dt;
  double wheel_base;
  double inv_wheel_base;
  double w_jerk, w_delta_rate, w_x, w_y, w_theta, w_v, w_a, w_delta;
  double abs_tol, rel_tol;
  double max_velocity, min_acc, max_acc, jerk_min, jerk_max, delta_min, delta_max;
  double delta_rate_min, delta_rate_max;
  double bar_r;        // 1 / t            (barrier_function.h:85)
  double bar_eps;      // epsilon
  double bar_inv_eps;  // 1 / epsilon
  double bar_half_r_inv_eps2;  // r / (2 eps^2)
  double bar_rlogeps;  // r * log(eps), evaluated once on the host in fp64
  double disc_off[kMaxDiscs];  // L*(j-0.5) - rf   (ilqr_optimizer.cc:564)
  double shrink_corridor;      // disc_radius + safe_margin   (cc:448)
  double shrink_lane;          // disc_radius                 (cc:463)
};

// TrackerConfig (planner_config.h:18-43): the alternative init guess of kernels_tracker.hip
struct TrackerParams {
  double weight_l, weight_theta, weight_delta, weight_delta_rate, preview_time;
  double weight_s, weight_v, weight_a, weight_j;
  double sim_dt, dt, tolerance;
  int max_num_iteration;
  int have_station;   // cstation was filled from the caller's stations (else: chord length, computed by the kernel)
};

struct DeviceState {
  int Bcap;  // arena capacity (slots): stride of every SLOT-indexed tensor
  int Pcap;  // stride of the PROBLEM-indexed tensors (hist, atrace): the capacity of the batch, whatever arena the
             // active problems currently live in (the finishing arena of solver.hip is smaller than the batch)
  int cmax;
  int nl, nr;  // lane segments
  int exact_ties;   // CILQR_OPT_EXACT_LANE_TIES: near-ties of the nearest-segment search decided by the reference's distances
  Params p;

  double2* X;      // [2][K][3][Bcap]
  double2* U;      // [2][N][Bcap]
  int* cur;        // [Bcap] which X/U buffer is the current iterate
  double2* goals;  // [K][3][Bcap]
  double2* coarse0;   // [2][Bcap] (x, y) (theta, v) of the coarse trajectory's first point (goals[0] is the start state)
  double* cstation;   // [K][Bcap] stations of the coarse trajectory (tracker init guess only)
  double* cor;     // [K][cmax][3][Bcap]
  int* ccnt;       // [K][Bcap]
  double* lanes;   // [nl+nr][kLaneFields]
  // uniform grid over the lane polylines: per cell the (conservative) set of segments that can be
  // the nearest one for a point of that cell, in ascending index order -> same result as the
  // reference's linear scan with ~2 distance tests instead of nl / nr
  unsigned char* lgrid;  // [2 sides][gnx*gny][kGridCellBytes]: count | up to 15 segment indices
  double gx0, gy0, ginv_h;
  int gnx, gny;
  // lin / term / gains are per-ITERATION scratch (written by the quadratisation, read by the backward pass, whose gains
  // the rollouts read): a slot's rows live at its POSITION in the iteration's active list (posn), not at the slot, so
  // that the three big streams of an iteration stay dense while the slots thin out between two re-packings
  // (dev_model.hpp: scratch_index).  posn == nullptr (the tail kernel's private view): position = slot.  The stage API works
  // on the main arena WITH its posn: cilqr_stage_load's k_load_goals writes the identity, and every solve clears the stage
  // flags (solver.hip: h->stage = 0), so a stage call can only follow a load -- it never sees a solve's positions.
  // Inside a solve: k_update writes the NEXT iteration's positions, so nothing launched after it in an iteration
  // (exports, re-packing) may call scratch_index for the current one -- none does.
  double2* lin;    // [N][17][Bcap]
  double2* term;   // [9][Bcap]
  double2* gains;  // [N][7][Bcap]
  int* posn;       // [Bcap] slot -> position in the current active list (k_load_goals: identity; k_update; k_compact: identity)
  double* dV;      // [2][Bcap]
  double* gnorm;   // [Bcap]
  double2* part;   // [K][3][Bcap]
  double* trial;   // [5][Bcap] cost components of the last evaluated trajectory
  double* hist;    // [max_iter+1][5][Bcap]

  // Slots vs problems.  A slot is where a problem's working set currently lives; once the active
  // set has halved, the survivors are re-packed into the first slots of the twin arena
  // (k_compact) so that every kernel keeps reading dense, coalesced rows.  pid[slot] is the
  // caller's problem index.  Bookkeeping that is touched once per iteration (hist, status, n_cost,
  // iter, n_iter_trajs) is indexed by PROBLEM and never moves; finished problems are exported the
  // moment they finish (k_export_done).
  int* pid;          // [Bcap] slot -> problem
  int* done_now;     // [Bcap] slot finished in this iteration's update

  // per-problem solver state (ilqr_optimizer.cc:180-199)
  double* lambda;
  double* dlambda;
  double* cost_old;
  double* dcost;     // of the accepted trial
  int* iter;
  int* status;
  int* n_cost;
  int* upd;          // is_forward_pass_updated
  int* acc_idx;      // accepted alpha index of this iteration, -1 = none yet, -2 = left the line search
  int* n_iter_trajs;
  int* emit;         // iterate to append to iter_trajs this iteration
  // [max_iter][Bcap] by PROBLEM: accepted alpha index of every iteration (-1 all rejected, -2 gradient-norm exit)
  signed char* atrace;

  // Candidate arena of the line search: rollouts and their per-knot cost partials, indexed [step size][...][list position]
  // with row stride spec_cap.  The arena holds spec_rows x capacity candidates ("cells"): all eleven step sizes of every slot
  // for small arenas, FOUR per slot for the big ones (kernels_search.hip, launch_linesearch: the pre-rolled rounds use rows
  // 0..3 at stride capacity; what they leave dead is then re-strided for the remaining step sizes of the problems that
  // rejected them all, or -- small active sets -- for all eleven of every active problem).  A launch sees ONE such view:
  // spec_cap and the three pointers are set per launch by the host (spec_view below), the kernels index as they always did.
  int spec_cap;
  int spec_rows;     // candidates per slot the allocation holds (4 or 11); the allocation's capacity is Bcap
  double2* Xs;       // [rows][K][3][spec_cap]
  double2* Us;       // [rows][N][spec_cap]
  double2* parts;    // [rows][K][3][spec_cap]
  double* spec_tot;  // [11][5][spec_cap] total cost of every candidate

  // work lists
  int* act;          // active slots
  int* act_next;
  int* pend;         // [kNumAlpha+1][Bcap] line-search pending lists (round r uses list r)
  int* counters;     // [0] = n_act_next, [1..kNumAlpha] = pending counts of rounds 1..10,
                     // [kCntActive] = active count of the iteration in flight
  // Non-null inside the solve loop: the host runs a couple of iterations ahead of the GPU and only
  // knows an upper bound of the active count; kernels clamp to the device-side value.
  const int* n_dev;
  // where k_update of the iteration in flight counts the survivors (the next iteration's n_dev), the
  // ring entry it clears for the iteration after that, and the host-visible copy of the count
  int* n_next;
  int* n_clear;
  int* h_count_dev;
};
constexpr int kCntActive = 32;   // [32..34]: ring of active counts, iteration it reads entry it % 3
constexpr int kCntTicket = 40;   // blocks of k_update that have finished

// ---- launchers (one per kernel family; all asynchronous on `st`) ----
struct ProblemView {  // device pointers to the problem-major inputs
  const double* start;
  const double* coarse;
  const double* station;   // [B][K] or nullptr
  const double* corridor;
  const int* ccount;
  int cmax_in;
};

void launch_load(const DeviceState& s, int B, const ProblemView& in, const double* lanes_raw,
                 hipStream_t st);
void launch_build_lane_grid(const DeviceState& s, hipStream_t st);
void launch_nearest_lane(const DeviceState& s, int n, const double* xy, int* left, int* right, int use_grid,
                         hipStream_t st);
void launch_init_guess(const DeviceState& s, int B, hipStream_t st);
void launch_init_guess_tracker(const DeviceState& s, const TrackerParams& tp, int B, hipStream_t st);
void launch_set_trajectory(const DeviceState& s, int B, const double* X, const double* U, hipStream_t st);
// cost of buffer (cur ^ cand) for the n listed slots -> trial[], no accept logic
void launch_cost_only(const DeviceState& s, const int* list, int n, int cand, hipStream_t st);
void launch_cost_knots(const DeviceState& s, const int* list, const int* n_ptr, int n_max, int n_grid,
                       int cand, int skip_done, hipStream_t st);
void launch_spec_cost(const DeviceState& s, const int* list, const int* n_ptr, int off, int n_max, int n_grid, int r0,
                      int sparse, hipStream_t st);
int spec_open_capacity(const DeviceState& s);   // most active problems an all-eleven-step-sizes pass can take (kernels_search.hip)
void launch_round_cost(const DeviceState& s, int r0, int group, int n_max, int n_grid, hipStream_t st);
void launch_init_cost_commit(const DeviceState& s, int n, hipStream_t st);
void launch_quadratize(const DeviceState& s, const int* list, int n, int only_upd, hipStream_t st);
void launch_backward(const DeviceState& s, const int* list, int n, const double* lambda_override,
                     int team_threshold, int wave_threshold, hipStream_t st, hipEvent_t ev_start = nullptr,
                     hipEvent_t ev_stop = nullptr);
void launch_forward(const DeviceState& s, const int* list, int n, double alpha, int skip_done,
                    hipStream_t st);
// the 11-round line search of one lockstep iteration (forward/cost/accept with compaction)
void launch_linesearch(const DeviceState& s, int n_act, int spec_threshold, int seq_rounds, int round_group,
                       hipStream_t st);
void launch_init_counters(const DeviceState& s, int first_n, hipStream_t st);
// kernels_tail.hip: one workgroup per problem finishes every problem of the active list (all remaining iterations)
size_t tail_workspace_bytes(const DeviceState& s);   // private arena of one problem
bool tail_supported(const DeviceState& s);           // the kernel's fixed LDS block fits what every device grants
void launch_tail(const DeviceState& s, void* workspace, int n_max, double* traj, double* iter_trajs,
                 int max_iter_trajs, int* max_iter_dev, hipStream_t st);
void launch_update(const DeviceState& s, int n_act, hipStream_t st);
// trajectories of the slots that finished in the last update -> traj[pid]
void launch_export_done(const DeviceState& s, int n_act, double* traj, hipStream_t st);
// survivors (next active list of `src`, n of them) -> slots 0..n-1 of `dst`
void launch_compact(const DeviceState& src, const DeviceState& dst, int n_max, hipStream_t st);
void launch_export_iter_traj(const DeviceState& s, const int* list, int n, double* iter_trajs,
                             int max_iter_trajs, hipStream_t st);
void launch_export_hist(const DeviceState& s, int B, double* cost_hist, int* n_cost, int* status,
                        int* n_iter, int* n_iter_trajs, signed char* alpha_trace, hipStream_t st);
// the LIVE Cost rows only, packed problem after problem: off [B + 1] = first row of every problem and the total (an
// exclusive prefix sum of n_cost, built here), rows [off[B]][5].  What a large batch in host memory downloads instead of
// the dense [B][max_iter + 1][5] array (24 MB instead of 527 MB on the bench workload).
void launch_export_hist_rows(const DeviceState& s, int B, long long* off, double* rows, hipStream_t st);
// corridor producer (kernels_corridor.hip)
constexpr int kCorMaxPts = 320;  // obstacle points of one knot + the 8 (24) box points
struct CorridorParams {
  double max_diff_x, max_diff_y, radius, max_axis_x, max_axis_y;   // planner_config.h:75-86
  int per_edge;   // box points per edge: 2 (both ends), or 6 with is_multiple_sample (corridor.cc:110-118)
};
void launch_build_corridors(int n, const CorridorParams& cp, const double* knots, const double* points,
                            const int* count, int pmax, double* corridor, int* ccount, int cmax, int* n_failed,
                            double* polygons, hipStream_t st);
void launch_device_math(int fn, int n, const double* in, double* out, hipStream_t st);
void launch_rollout(const Params& p, int B, const double* x0, const double* U, double* X, hipStream_t st);
// stage_read helpers: gather a batch-fastest tensor into problem-major order
void launch_gather_pairs(const double2* src, int rows_pairs, int Bcap, int B, double* dst,
                         int dst_stride, int dst_off, hipStream_t st);
void launch_gather_scalar(const double* src, int rows, int Bcap, int B, double* dst, int dst_stride,
                          int dst_off, hipStream_t st);
void launch_expand(const DeviceState& s, int B, int tensor, double* dst, hipStream_t st);
void launch_gather_xu(const DeviceState& s, int B, int cand, double* X, double* U, hipStream_t st);

}  // namespace cilqr
