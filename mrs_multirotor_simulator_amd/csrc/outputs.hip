// outputs.hip — per-UAV publisher payloads derived on the device and packed for ONE device-to-host copy.
// Replaces the per-UAV host work of UavSystemRos::publishOdometry/IMU/Rangefinder (src/uav_system_ros.cpp:342-431)
// and MultirotorSimulator::publishPoses (src/multirotor_simulator.cpp:365-389): state never leaves HBM column by column.
// HBM-bound gather: reads 22 doubles + flags per UAV (coalesced SoA columns), writes one 136-B record.
#include <hip/hip_runtime.h>

#include "../../include/mrs_swarm.h"
#include "swarm_layout.h"

namespace {

// Eigen::Quaterniond(Matrix3d) (what mrs_lib::AttitudeConverter(R) stores): Eigen/src/Geometry/Quaternion.h,
// quaternionbase_assign_impl<Other,3,3>.  R row-major; q = {x, y, z, w}.
__device__ __forceinline__ void quat_from_matrix(const double m[9], double q[4]) {
  double t = (m[0] + m[4]) + m[8];
  if (t > 0) {
    t    = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t    = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    // i = argmax of the diagonal with Eigen's tie rules; written without dynamic register indexing
    const bool i1 = m[4] > m[0];
    const double mi1 = i1 ? m[4] : m[0];
    const bool i2 = m[8] > mi1;
    if (i2) {  // i=2, j=0, k=1
      t    = sqrt(m[8] - m[0] - m[4] + 1.0);
      q[2] = 0.5 * t;
      t    = 0.5 / t;
      q[3] = (m[3] - m[1]) * t;  // (m(k,j) - m(j,k)) = m(1,0) - m(0,1)
      q[0] = (m[2] + m[6]) * t;  // (m(j,i) + m(i,j)) = m(0,2) + m(2,0)
      q[1] = (m[5] + m[7]) * t;  // (m(k,i) + m(i,k)) = m(1,2) + m(2,1)
    } else if (i1) {  // i=1, j=2, k=0
      t    = sqrt(m[4] - m[8] - m[0] + 1.0);
      q[1] = 0.5 * t;
      t    = 0.5 / t;
      q[3] = (m[2] - m[6]) * t;  // m(0,2) - m(2,0)
      q[2] = (m[7] + m[5]) * t;  // m(2,1) + m(1,2)
      q[0] = (m[1] + m[3]) * t;  // m(0,1) + m(1,0)
    } else {  // i=0, j=1, k=2
      t    = sqrt(m[0] - m[4] - m[8] + 1.0);
      q[0] = 0.5 * t;
      t    = 0.5 / t;
      q[3] = (m[7] - m[5]) * t;  // m(2,1) - m(1,2)
      q[1] = (m[3] + m[1]) * t;  // m(1,0) + m(0,1)
      q[2] = (m[6] + m[2]) * t;  // m(2,0) + m(0,2)
    }
  }
}

__global__ void __launch_bounds__(256) k_pack_outputs(SwarmDev sw, int first, int count, mrs_uav_output_t* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int    i  = first + k;
  const size_t np = (size_t)sw.npad;
#define LD(f) sw.S[(f) * np + i]
  double x[3], v[3], R[9], w[3], imu[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    x[c] = LD(F_X + c); v[c] = LD(F_V + c); w[c] = LD(F_W + c); imu[c] = LD(F_IMU + c);
  }
#pragma unroll
  for (int c = 0; c < 9; c++) R[c] = LD(F_R + c);
#undef LD
  const double ground_z = sw.T[sw.F[i] >> FLAG_TYPE_SHIFT].ground_z;
  mrs_uav_output_t o;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    o.position[c]            = x[c];
    o.angular_velocity[c]    = w[c];
    o.linear_acceleration[c] = imu[c];
    o.velocity_body[c]       = (R[c] * v[0] + R[3 + c] * v[1]) + R[6 + c] * v[2];  // R^T v, src/uav_system_ros.cpp:356
  }
  quat_from_matrix(R, o.orientation);
  // publishRangefinder, :403-419: dot(-body_z, (0,0,-1)) = ((-bz0)*0 + (-bz1)*0) + (-bz2)*(-1)
  const double bz2  = R[8];
  const double dot  = ((-R[2]) * 0.0 + (-R[5]) * 0.0) + (-bz2) * (-1.0);
  const double tilt = acos(dot);
  double       range = 1.7976931348623157e308;
  if (bz2 > 0) range = (x[2] - ground_z) / cos(tilt) + 0.01;
  if (range > 40.0) range = 41.0;
  o.range = range;
  out[k]  = o;
}

// mrs_lib::AttitudeConverter(R).getHeading(): q = Eigen::Quaterniond(R); heading = atan2 of the first column of
// tf2::Matrix3x3::setRotation(q) (tf2::Transform(q) * (1,0,0))
__device__ __forceinline__ double heading_of(const double R[9]) {
  double q[4];
  quat_from_matrix(R, q);
  const double d  = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  const double sc = 2.0 / d;
  const double ys = q[1] * sc, zs = q[2] * sc;
  const double wz = q[3] * zs, xy = q[0] * ys, yy = q[1] * ys, zz = q[2] * zs;
  return atan2(xy + wz, 1.0 - (yy + zz));
}

// UavSystemRos::timeoutInput (src/uav_system_ros.cpp:474-647) for one UAV per lane: the command columns are overwritten
// with the safe command of the UAV's current input mode; state stays on the device.
__global__ void __launch_bounds__(256) k_timeout_input(SwarmDev sw, int first, int count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int      i    = first + k;
  const size_t   np   = (size_t)sw.npad;
  const uint32_t fl   = sw.F[i];
  const int      mode = (int)((fl & FLAG_MODE_MASK) >> FLAG_MODE_SHIFT);
#define CMD(q) sw.S[(F_CMD + (q)) * np + i]
  double R[9];
#pragma unroll
  for (int c = 0; c < 9; c++) R[c] = sw.S[(F_R + c) * np + i];
  switch (mode) {
    case MRS_POSITION_CMD:
      CMD(0) = sw.S[(F_X + 0) * np + i]; CMD(1) = sw.S[(F_X + 1) * np + i]; CMD(2) = sw.S[(F_X + 2) * np + i];
      CMD(3) = heading_of(R);
      break;
    case MRS_VELOCITY_HDG_CMD:
    case MRS_ACCELERATION_HDG_CMD:
      CMD(0) = 0.0; CMD(1) = 0.0; CMD(2) = 0.0;
      CMD(3) = heading_of(R);
      break;
    case MRS_VELOCITY_HDG_RATE_CMD:
    case MRS_ACCELERATION_HDG_RATE_CMD:
    case MRS_ATTITUDE_RATE_CMD:
    case MRS_CONTROL_GROUP_CMD:
      CMD(0) = 0.0; CMD(1) = 0.0; CMD(2) = 0.0; CMD(3) = 0.0;
      break;
    case MRS_ATTITUDE_CMD: {
      // mrs_lib::AttitudeConverter(0, 0, heading): tf2 setRPY, then Eigen Quaternion::toRotationMatrix
      const double h  = heading_of(R);
      const double hy = h * 0.5, hp = 0.0 * 0.5, hr = 0.0 * 0.5;
      const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
      const double x = sr * cp * cy - cr * sp * sy, y = cr * sp * cy + sr * cp * sy, z = cr * cp * sy - sr * sp * cy,
                   w = cr * cp * cy + sr * sp * sy;
      const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
      const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                   tzz = tz * z;
      CMD(0) = 1.0 - (tyy + tzz); CMD(1) = txy - twz;         CMD(2) = txz + twy;
      CMD(3) = txy + twz;         CMD(4) = 1.0 - (txx + tzz); CMD(5) = tyz - twx;
      CMD(6) = txz - twy;         CMD(7) = tyz + twx;         CMD(8) = 1.0 - (txx + tyy);
      CMD(9) = 0.0;
      break;
    }
    case MRS_TILT_HDG_RATE_CMD:
      CMD(0) = 0.0; CMD(1) = 0.0; CMD(2) = 1.0; CMD(3) = 0.0; CMD(4) = 0.0;
      break;
    case MRS_ACTUATOR_CMD:
#pragma unroll
      for (int m = 0; m < MRS_MAXM; m++) CMD(m) = 0.0;
      break;
    default: break;  // INPUT_UNKNOWN stays INPUT_UNKNOWN
  }
#undef CMD
}

}  // namespace

// Staged command upload: `rows` holds count x stride doubles in the caller's row layout (one UAV per row, as setInput receives
// them); element j of every row goes to column `base + j`.  One coalesced read of the rows, `width` coalesced column writes.
__global__ void __launch_bounds__(256) k_unpack_rows(SwarmDev sw, const double* rows, int stride, int width, int base, int first, int count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const size_t np = (size_t)sw.npad;
  for (int j = 0; j < width; j++) sw.S[(size_t)(base + j) * np + first + k] = rows[(size_t)k * stride + j];
}

extern "C" hipError_t mrs_launch_unpack_rows(SwarmDev sw, const double* rows, int stride, int width, int base, int first, int count,
                                             hipStream_t st) {
  if (count <= 0 || width <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpack_rows, dim3((count + 255) / 256), dim3(256), 0, st, sw, rows, stride, width, base, first, count);
  return hipGetLastError();
}

extern "C" hipError_t mrs_launch_timeout_input(SwarmDev sw, int first, int count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_timeout_input, dim3((count + 255) / 256), dim3(256), 0, st, sw, first, count);
  return hipGetLastError();
}

// MultirotorModel::State + IMU + crash flag of UAVs [first, first + count) as packed records (mrs_uav_state_t)
namespace {
__global__ void __launch_bounds__(256) k_pack_states(SwarmDev sw, int first, int count, mrs_uav_state_t* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int      i  = first + k;
  const size_t   np = (size_t)sw.npad;
  const uint32_t fl = sw.F[i];
  mrs_uav_state_t o;
#define LD(f) sw.S[(size_t)(f) * np + i]
#pragma unroll
  for (int c = 0; c < 3; c++) {
    o.x[c] = LD(F_X + c); o.v[c] = LD(F_V + c); o.omega[c] = LD(F_W + c); o.imu_acceleration[c] = LD(F_IMU + c);
    o.v_prev[c] = (fl & FLAG_VPREV_SPLIT) ? LD(F_VPREV + c) : o.v[c];  // v_prev == v unless setState changed v since the last step
  }
#pragma unroll
  for (int c = 0; c < 9; c++) o.R[c] = LD(F_R + c);
  const int nm = sw.T[fl >> FLAG_TYPE_SHIFT].n_motors;
#pragma unroll
  for (int m = 0; m < MRS_MAX_MOTORS; m++) o.motor_rpm[m] = m < nm ? LD(F_RPM + m) : 0.0;
#undef LD
  o.crashed  = (fl & FLAG_CRASHED) ? 1 : 0;
  o.n_motors = nm;
  out[k] = o;
}
}  // namespace
extern "C" hipError_t mrs_launch_pack_states(SwarmDev sw, int first, int count, mrs_uav_state_t* dev_out, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_states, dim3((count + 255) / 256), dim3(256), 0, st, sw, first, count, dev_out);
  return hipGetLastError();
}

extern "C" hipError_t mrs_launch_pack_outputs(SwarmDev sw, int first, int count, mrs_uav_output_t* dev_out, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_outputs, dim3((count + 255) / 256), dim3(256), 0, st, sw, first, count, dev_out);
  return hipGetLastError();
}
