/* cl_compat.h — the handful of OpenCL names the reference's CALLERS use
 * (test_cfft.cpp:24-40, csound/opcode.cpp:50-64), mapped onto HIP device ordinals.
 *
 * Not an OpenCL implementation: only device enumeration, the device name query,
 * the scalar typedefs and the status codes — and, for SUBCLASSES of the reference's
 * classes (they reach the protected members commands / data1 / data2 / w / b,
 * cl_fft.h:35-44, with clEnqueueWriteBuffer / clEnqueueReadBuffer / clFinish as
 * cl_fft.cpp:155-158 does), those three calls on a hipStream_t and device pointers.  A cl_device_id is an opaque handle
 * that encodes "HIP ordinal + 1"; it is only meaningful to the classes in
 * cl_fft.h / cl_conv.h / cl_dconv.h of this package.
 *
 * Do not include <CL/opencl.h> in the same translation unit: handles returned by
 * a real OpenCL runtime cannot be mapped to HIP devices.  (If the real header
 * was included first, the types are reused and only the two shim functions are
 * renamed out of the way; define CLFA_NO_CL_SHIM to omit them entirely.)
 */
#ifndef CLFFT_AMD_CL_COMPAT_H
#define CLFFT_AMD_CL_COMPAT_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../clfft_amd.h"

#ifndef __OPENCL_CL_H
typedef int32_t cl_int;
typedef uint32_t cl_uint;
typedef uint64_t cl_ulong;
typedef cl_ulong cl_bitfield;
typedef cl_bitfield cl_device_type;
typedef cl_uint cl_device_info;
typedef struct _cl_platform_id *cl_platform_id;
typedef struct _cl_device_id *cl_device_id;
typedef float cl_float;
typedef struct { float s[2]; } cl_float2;
typedef cl_uint cl_bool;
typedef void *cl_mem;             /* a device pointer */
typedef void *cl_command_queue;   /* a hipStream_t */
typedef struct _cl_event *cl_event;
#define CL_TRUE 1
#define CL_FALSE 0

#define CL_SUCCESS 0
#define CL_DEVICE_NOT_FOUND -1
#define CL_DEVICE_NOT_AVAILABLE -2
#define CL_COMPILER_NOT_AVAILABLE -3
#define CL_MEM_OBJECT_ALLOCATION_FAILURE -4
#define CL_OUT_OF_RESOURCES -5
#define CL_OUT_OF_HOST_MEMORY -6
#define CL_INVALID_VALUE -30
#define CL_INVALID_DEVICE -33
#define CL_DEVICE_TYPE_DEFAULT (1 << 0)
#define CL_DEVICE_TYPE_CPU (1 << 1)
#define CL_DEVICE_TYPE_GPU (1 << 2)
#define CL_DEVICE_TYPE_ALL 0xFFFFFFFF
#define CL_DEVICE_NAME 0x102B
#define CLFA_DEFINED_CL_TYPES 1
#endif

/* handle <-> HIP ordinal */
static inline cl_device_id clfa_device_handle(int ordinal) { return (cl_device_id)(intptr_t)(ordinal + 1); }
static inline int clfa_device_ordinal(cl_device_id id) { return (int)(intptr_t)id - 1; }

#ifndef CLFA_NO_CL_SHIM
/* clGetDeviceIDs(NULL, CL_DEVICE_TYPE_ALL, 32, ids, &num) as at test_cfft.cpp:31 */
static inline cl_int clfa_clGetDeviceIDs(cl_platform_id platform, cl_device_type type, cl_uint num_entries,
                                          cl_device_id *devices, cl_uint *num_devices) {
  (void)platform;
  if (type == CL_DEVICE_TYPE_CPU) return CL_DEVICE_NOT_FOUND; /* there is no CPU path */
  int n = 0;
  int e = clfa_device_count(&n);
  if (e != 0 || n <= 0) return CL_DEVICE_NOT_FOUND;
  if (num_devices) *num_devices = (cl_uint)n;
  if (devices)
    for (cl_uint i = 0; i < num_entries && i < (cl_uint)n; i++) devices[i] = clfa_device_handle((int)i);
  return CL_SUCCESS;
}
/* clGetDeviceInfo(id, CL_DEVICE_NAME, 128, name, NULL) as at test_cfft.cpp:37 */
static inline cl_int clfa_clGetDeviceInfo(cl_device_id id, cl_device_info what, size_t size, void *value,
                                           size_t *size_ret) {
  if (what != CL_DEVICE_NAME || !value || size == 0) return CL_INVALID_VALUE;
  int e = clfa_device_name(clfa_device_ordinal(id), (char *)value, size);
  if (e == 0 && size_ret) *size_ret = strlen((const char *)value) + 1;
  return e;
}
#ifdef CLFA_DEFINED_CL_TYPES
/* cl_fft.cpp:155 / :158 / cl_conv.cpp:38-41: copies on the object's queue, events ignored (none are ever passed) */
static inline cl_int clfa_clEnqueueWriteBuffer(cl_command_queue q, cl_mem buf, cl_bool blocking, size_t off, size_t bytes,
                                                const void *ptr, cl_uint nev, const cl_event *wait, cl_event *ev) {
  (void)nev; (void)wait; (void)ev;
  return clfa_copy_to_device(q, (char *)buf + off, ptr, bytes, (int)blocking);
}
static inline cl_int clfa_clEnqueueReadBuffer(cl_command_queue q, cl_mem buf, cl_bool blocking, size_t off, size_t bytes,
                                               void *ptr, cl_uint nev, const cl_event *wait, cl_event *ev) {
  (void)nev; (void)wait; (void)ev;
  return clfa_copy_from_device(q, ptr, (const char *)buf + off, bytes, (int)blocking);
}
static inline cl_int clfa_clFinish(cl_command_queue q) { return clfa_stream_synchronize(q); }
#define clGetDeviceIDs clfa_clGetDeviceIDs
#define clGetDeviceInfo clfa_clGetDeviceInfo
#define clEnqueueWriteBuffer clfa_clEnqueueWriteBuffer
#define clEnqueueReadBuffer clfa_clEnqueueReadBuffer
#define clFinish clfa_clFinish
#endif
#endif /* CLFA_NO_CL_SHIM */

#endif
