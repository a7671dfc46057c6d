// swarm_layout.h — device data layout shared by the host library and the kernels.
//
// State lives in HBM as field-major SoA FP64: S[field * npad + uav] (npad = n rounded up to 64), so
// lane l of a wavefront reads uav base+l of every field with one coalesced 512-B request.  Per-UAV
// flags are one u32.  Per-airframe constants ("types") sit in a small table that the step kernel
// reads through the scalar cache (one type per wavefront iteration — see step_device.inc).
#pragma once
#include <stdint.h>

#define MRS_MAXM 8

// ---- SoA field indices (doubles) ----
enum {
  F_X     = 0,   // 3  position                      MultirotorModel::State::x      multirotor_model.hpp:92
  F_V     = 3,   // 3  velocity                                     ::v                                   :93
  F_VPREV = 6,   // 3                                               ::v_prev                              :94
  F_R     = 9,   // 9  rotation matrix, row-major                   ::R                                   :95
  F_W     = 18,  // 3  body rates                                   ::omega                               :96
  F_RPM   = 21,  // 8  motor rpm (first n_motors used)              ::motor_rpm                           :97
  F_IMU   = 29,  // 3  imu_acceleration_                                                                  :139
  F_FEXT  = 32,  // 3  external_force_                                                                    :142
  F_INITZ = 35,  // 1  _initial_pos_(2)                                                                   :145
  F_PID   = 36,  // 24 {position,velocity,attitude,rate} x {x,y,z} x {last_error_, integral_}  pid.hpp:20-21
  F_CMD   = 60,  // 10 payload of the active input (layout per mode, see mrs_swarm.h)   uav_system.hpp:99-108
  F_FF    = 70,  // 16 four feed-forward slots x {vec3, heading(_rate)}                 uav_system.hpp:112-115
  F_COUNT = 86
};

// ---- per-UAV flag word ----
#define FLAG_CRASHED   0x1u        // UavSystem::crashed_                    uav_system.hpp:80
#define FLAG_TAKEOFF   0x2u        // params_.takeoff_patch_enabled (mutated by step())  multirotor_model.hpp:264-276
#define FLAG_MODE_SHIFT 2          // 4 bits: UavSystem::active_input_       uav_system.hpp:95
#define FLAG_MODE_MASK (0xFu << FLAG_MODE_SHIFT)
#define FLAG_FF_SHIFT  6           // 4 bits: which std::optional feed-forwards hold a value
#define FLAG_FF_MASK   (0xFu << FLAG_FF_SHIFT)
#define FLAG_VPREV_SPLIT 0x400u   // v_prev differs from v: the F_VPREV column is authoritative.  After every step v_prev == v
                                  // (multirotor_model.hpp:281), so the step kernel neither reads nor writes F_VPREV unless
                                  // MultirotorModel::setState changed v in between (:424-433 leaves v_prev alone)
#define FLAG_HOLD       0x800u     // UavSystemRos: the model is not iterated (no input yet / input timed out, and
                                  // iterate_without_input == false)            src/uav_system_ros.cpp:265
#define FLAG_TYPE_SHIFT 16         // 16 bits: index into the type table
#define MRS_MAX_TYPES  65536

// ---- per-type constants (device copy; everything the kernels need, pre-derived on the host with the
//      reference's own operation order so the values are bit-identical to computing them per step) ----
struct TypeParams {
  int32_t n_motors, ground_enabled, desaturation, _pad;
  double  g, mass, inv_mass, min_rpm, max_rpm;
  double  kf_n;       // kf * n_motors                                  acceleration_controller.hpp:92
  double  resist_k;   // ((air_resistance_coeff * M_PI) * arm) * arm    multirotor_model.hpp:337 (prefix of the product chain)
  double  hover_thr;  // 0.90 * sqrt((mass*g)/(n_motors*kf))            multirotor_model.hpp:266-267
  double  ground_z;
  double  filt_c;     // exp(-dt/motor_time_constant) for the dt of the current launch   :244
  double  filt_1mc;   // 1.0 - filt_c
  double  tau;        // motor_time_constant
  double  inv_kf_n, inv_rpm_range;  // FAST flavour: 1/(kf*n_motors), 1/(max_rpm-min_rpm)
  double  arm_length, prop_radius;  // collision criterion             src/multirotor_simulator.cpp:342
  double  J[9], Jinv[9];            // Jinv = Eigen 3x3 cofactor inverse of J     multirotor_model.hpp:350
  double  alloc[4 * MRS_MAXM];      // torque/thrust allocation, row-major 4 x 8  :334
  double  alloc_inv[MRS_MAXM * 4];  // Mixer::allocation_matrix_inv_, n x 4       mixer.hpp:72-101
  double  pos_kp, pos_kd, pos_ki, pos_sat;                 // position_controller.hpp:92-103
  double  vel_kp, vel_kd, vel_ki, vel_sat;                 // velocity_controller.hpp:108-119
  double  att_kp, att_kd, att_ki, att_sat_rp, att_sat_yaw; // attitude_controller.hpp:160-171
  double  rate_kp[3], rate_kd[3], rate_ki[3];              // gains * J(i,i), rate_controller.hpp:56-65
  // displacement bound of the sharded collision tick ("prediction"): |acceleration| over the coming steps is at most
  //   pred_a0 + pred_thr * |allocation*rpm^2 thrust of this step| + (listed partners) * |rebounce| + pred_drag * speed^2
  double  pred_a0;    // g + 1.5 * (sum_m alloc[3][m]) * max_rpm^2 / mass; +inf when the bound cannot be given (a negative thrust column)
  double  pred_thr;   // 1.5 / mass (times the thrust of the current step: covers motor speeds beyond max_rpm — set by the host, or a lowered max_rpm)
  double  pred_drag;  // |resist_k| / mass
};

// ---- device view of a swarm ----
struct PosRecord;
struct SwarmDev {
  double*             S;      // F_COUNT x npad
  uint32_t*           F;      // npad
  const TypeParams*   T;      // type table
  unsigned long long* diag;   // 4 counters (mrs_diag_t order)
  const uint32_t*     BT;     // per 64-UAV block: airframe type (0xFFFF = mixed types) | n_motors << 16
  const int32_t*      MB;     // indices of the mixed blocks (n_mixed entries)
  int32_t             n, npad, n_mixed;
  uint32_t            opts;   // bit 0: some UAV may carry a non-zero external force (else the F_FEXT columns are not read)
  // skin test of the collision pass's neighbour lists (collide.hip); vl_flag == nullptr: no lists are live
  const PosRecord*    vl_rec;   // per-UAV record holding the position at the last list rebuild
  uint32_t*           vl_flag;  // set to 1 when some UAV is farther than sqrt(vl_lim2) from that position
  double              vl_lim2;
  int32_t             blk0;     // step kernels: first 64-UAV block of this launch (a step may be split over two streams)
  int32_t             fast;     // MRS_ARITH_FAST swarm: the stand-alone collision passes use the FAST force expression too (collide_device.inc)
};

// 48-byte record exchanged for the collision pass (single- and multi-GPU): everything
// MultirotorSimulator::handleCollisions reads of the partner UAV (src/multirotor_simulator.cpp:339-350)
struct PosRecord {
  double x, y, z;
  double mass, arm_length, prop_radius;
};

// ---- fused step + collision evaluation (step_device.inc *_coll kernels; buffers owned by collide.hip) ----
// A collision tick between two neighbour searches is evaluated by the NEXT step kernel: its prologue reads the positions of
// the listed partners as they were after the previous step, forms the force / crash flag handleCollisions would have
// latched (src/multirotor_simulator.cpp:321-358) and the step consumes it from registers.  Positions are double-buffered in
// their own 32-B records so that the partners' values do not change under a running launch.
struct Pos4 {
  double x, y, z, w;  // position records: w = airframe type.  The HEADER record of an export block is read as 32-bit words instead:
                      // word 0 = smallest stall index the rank knows of, word 1 = smallest warning index (0: none) — MRS_HDR_*
};
#define MRS_HDR_STALL 0
#define MRS_HDR_WARN  1
#define MRS_HDR_ERROR 2  // the rank's CTL_ERROR bits: every rank's call fails when any rank's kernels reported an error
// one record of the halo exchange of a search tick (collide.hip mrs_collide_halo_*): a PosRecord + the UAV's index on its rank
struct HaloEntry {
  double             x, y, z, mass, arm_length, prop_radius;
  unsigned long long j, pad;  // (header entry of a block: j = entries that follow, pad = flags)
};
struct PartnerConst {
  double mass, arm_length, prop_radius, _pad;
};
#define MRS_NBR_FOREIGN 0x80000000u  // neighbour-list entry: index into the gathered export buffer instead of a local UAV index
#define MRS_NO_SLOT     0xFFFFFFFFu

// control words of the list state machine (device copy `ctl`, mirrored to pinned host memory `hostw` for the first two)
enum {
  CTL_STALL = 0,     // 0, or the tick index of the launch after which the lists stopped being usable (every later launch is a no-op)
  CTL_PROGRESS = 1,  // tick index of the last launch that started
  CTL_OVERFLOW = 2,  // UAVs with more than LIST_CAP listed neighbours at the last search
  CTL_BADSLOT = 3,   // export-set translation: foreign neighbours that their owner does not export (must stay 0: the relation is symmetric)
  CTL_EXPORTS = 4,   // export-set search: number of own UAVs that some other rank lists
  CTL_WARN = 5,      // tick index of the last launch in which some UAV was beyond the WARNING part of its skin: the host schedules the
                     // next search ahead of time, in stream order, instead of waiting for the stall (host mirror only)
  // split sharded ticks (interior / boundary launches on two streams): the boundary chain mirrors what it knows into host
  // words of its own — one writer per word at any time; the host takes the smaller non-zero of a pair
  CTL_STALL2 = 6,
  CTL_WARN2 = 7,
  CTL_ERROR = 8,     // bit 0: a bounded in-kernel wait ran out; bit 1: a UAV left its skin without the displacement bound announcing it;
                     // bits 8-9: the same, reported by SOME rank of the sharded swarm (folded from the export headers, MRS_HDR_ERROR)
  CTL_I_STARTED = 9, // tick index of the last interior launch that has started (so the interior launch before it is complete)
  CTL_NBND = 11,     // 64-UAV blocks of this rank that hold a boundary UAV (set by the search)
  CTL_NL1 = 12,      // interior blocks that list a UAV of a boundary block (MRS_BLK_LAYER1; set by the search, sent to the host with its head words)
  CTL_PRED = 13,     // set by the search when some own UAV may leave its skin within MRS_PRED_HORIZON steps (the ticks after the search are then serial)
  CTL_WORDS = 16
};
// class of a 64-UAV block in a split sharded tick (set by every search from the neighbour lists)
#define MRS_BLK_BOUNDARY 1u  // some UAV of the block lists a foreign UAV: the block is stepped by the boundary launch
#define MRS_BLK_LAYER1   2u  // interior block, some UAV of it lists a UAV of a boundary block: waits for that block's epoch word
#define MRS_PRED_HORIZON 4u  // steps by which "may leave its skin" is announced ahead (why 4: DESIGN §5)
// peer-window exchange (collide.hip k_peer_allgather): the windows of all ranks as this process addresses them
#define MRS_MAX_PEERS 64
struct MrsPeerWindows {
  void* win[MRS_MAX_PEERS];
};
enum { MRS_PART_FULL = 0, MRS_PART_INTERIOR = 1, MRS_PART_BOUNDARY = 2 };

struct CollDev {
  const uint32_t*     nbr;      // [LIST_CAP][n]: row k, UAV i at nbr[k * n + i]; ascending partner order
  const uint32_t*     nbr_cnt;  // [n]
  const PosRecord*    rec;      // [n] records of the last search: skin-test reference position + airframe constants of local partners
  const Pos4*         p_in;     // [n] positions after the previous step
  Pos4*               p_out;    // [n] positions after this step
  uint32_t*           ctl;      // CTL_WORDS control words
  volatile uint32_t*  hostw;    // pinned host mirror of ctl[CTL_STALL], ctl[CTL_PROGRESS]
  // export-set exchange (world > 1): every rank's block of the gathered buffer is [header][cap] Pos4 records
  const Pos4*         g_pos;    // [world * (1 + cap)] gathered export positions of the previous tick
  const PartnerConst* g_const;  // [world * (1 + cap)] airframe constants of the exported UAVs (fixed between searches)
  Pos4*               send;     // [1 + cap] this rank's block of the next all-gather
  const uint32_t*     exp_slot; // [n] export slot of every own UAV (MRS_NO_SLOT: nobody else lists it)
  double              rebounce, lim2, lim2_warn;
  uint32_t            tau;      // tick index of this launch (1, 2, ... since the host last drained the stream)
  int32_t             n, eval, crash, world, block;  // block = 1 + cap
  int32_t             write_force, part;             // latch the evaluated force in the F_ext columns as well; MRS_PART_*
  // split sharded ticks: block classes, the boundary launch's block list, per-block epoch words (tick index of the last launch that
  // finished the block), and the displacement bound's step-dependent factors
  const uint32_t*     blk_class;  // [blocks]
  const uint32_t*     blk_list;   // [blocks]: the boundary blocks from the front, the layer-1 blocks from the back
  uint32_t*           epoch;      // [blocks]
  uint32_t            n_bnd, n_l1;  // n_l1 > 0 (interior launch): the grid is n_l1 + blocks, its first n_l1 blocks take the layer-1 blocks of the list
  double              pred_hdt;   // horizon * dt; < 0: nothing is announced (serial protocol, MRS_SHARD_SPLIT=0: every launch tests exactly)
  double              pred_lim;   // sqrt(lim2)
};
