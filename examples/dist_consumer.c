/* A non-Python consumer of the several-GPU entry point: one rank of a Pr x Pc process grid evaluating
 * the block-cyclic GP log marginal likelihood through gpn_dist_lml_forward (include/gpnative.h), with
 * its own RCCL communicators wrapped by libgpnative_rccl.so.
 *
 *   gcc -std=c99 examples/dist_consumer.c -Iinclude -I/opt/rocm/include -Lgptorch_amd/lib -lgpnative \
 *       -lgpnative_rccl -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,$PWD/gptorch_amd/lib \
 *       -Wl,-rpath,/opt/rocm/lib -lm -o build/dist_consumer
 *   build/dist_consumer <n> <d> <tile> [<rank> <Pr> <Pc> <id-file> [<nonce>]]
 *
 * With no grid arguments it runs the 1 x 1 grid on one GPU with the collectives FORCED through RCCL
 * (single-member communicators), i.e. every ncclBroadcast / ncclAllReduce of a real run.  With a
 * grid, rank 0 publishes the ncclUniqueId at <id-file> (written aside and renamed into place, tagged with the
 * launcher's <nonce>) and the other ranks wait for THIS run's file (one process per GPU, HIP device = rank).
 * GPN_DIST_SCHEDULE=mesh selects the point-to-point panel exchange of the adapter (GPN_DIST_MESH_EXCHANGE).  Prints "lml=<value> info=<value>"; the inputs are those of
 * gptorch_amd/rng.py, so the value is the one the Python shell gives
 * (tests/test_gpu_parity.py::test_c_dist_consumer_runs). */
#define __HIP_PLATFORM_AMD__ 1
#define _GNU_SOURCE 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "gpnative.h"

#define CHECK(e) do { hipError_t _s = (e); if (_s != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_s)); return 2; } } while (0)
#define GPN(e) do { int _s = (e); if (_s != 0) { fprintf(stderr, "%s -> %d %s\n", #e, _s, gpn_last_hip_error()); return 3; } } while (0)
#define NCCL(e) do { ncclResult_t _s = (e); if (_s != ncclSuccess) { fprintf(stderr, "%s: %s\n", #e, ncclGetErrorString(_s)); return 4; } } while (0)

static uint64_t sm_state;
static uint64_t splitmix64(void) {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void normals(uint64_t seed, double* out, int64_t count) {   /* gptorch_amd/rng.py */
  sm_state = seed;
  for (int64_t i = 0; i < count; ++i) {
    const double u1 = ((double)(splitmix64() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(splitmix64() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    out[i] = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2048;
  const int d = argc > 2 ? atoi(argv[2]) : 8;
  const int64_t tile = argc > 3 ? atoll(argv[3]) : 512;
  const int grid = argc > 7;
  const int rank = grid ? atoi(argv[4]) : 0, pr = grid ? atoi(argv[5]) : 1, pc = grid ? atoi(argv[6]) : 1;
  const int dy = 1;
  double *x = malloc(sizeof(double) * n * d), *eps = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n);
  normals(0, x, n * d);                                   /* rng.make_regression(n, d, 1, seed=0) */
  normals(1, eps, n);
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += x[i * d + c];
    y[i] = sin(s) + 0.1 * eps[i];
  }
  const double theta[3] = {1.0, sqrt((double)d), 1e-2};   /* variance, length scale, noise */

  CHECK(hipSetDevice(grid ? rank : 0));
  /* communicators: world, then split by grid coordinate */
  ncclUniqueId id;
  /* The id travels through a file: rank 0 writes [nonce | id] to "<idfile>.tmp" and rename()s it into place (readers
   * never see a partial write); the other ranks poll until the file carries THIS run's nonce (argv[8], any string the
   * launcher makes up, e.g. its pid -- a file left behind by an earlier run is ignored, not joined). */
  char nonce[32] = {0}, seen[32];
  if (argc > 8) strncpy(nonce, argv[8], sizeof(nonce) - 1);
  if (rank == 0) {
    NCCL(ncclGetUniqueId(&id));
    if (grid) {
      char tmp[4096];
      snprintf(tmp, sizeof(tmp), "%s.tmp", argv[7]);
      remove(argv[7]);
      FILE* f = fopen(tmp, "wb");
      if (!f || fwrite(nonce, sizeof(nonce), 1, f) != 1 || fwrite(&id, sizeof(id), 1, f) != 1 || fclose(f) != 0 || rename(tmp, argv[7]) != 0) {
        fprintf(stderr, "cannot publish the unique id at %s\n", argv[7]);
        return 5;
      }
    }
  } else {
    int got = 0;
    for (int tries = 0; tries < 600 && !got; ++tries) {
      FILE* f = fopen(argv[7], "rb");
      if (f) {
        got = fread(seen, sizeof(seen), 1, f) == 1 && fread(&id, sizeof(id), 1, f) == 1 && memcmp(seen, nonce, sizeof(nonce)) == 0;
        fclose(f);
      }
      if (!got) usleep(100000);
    }
    if (!got) { fprintf(stderr, "no unique id for this run at %s\n", argv[7]); return 5; }
  }
  ncclComm_t world, row, col;
  NCCL(ncclCommInitRank(&world, pr * pc, id, rank));
  NCCL(ncclCommSplit(world, rank / pc, rank % pc, &row, NULL));     /* rank in `row` = process-column index */
  NCCL(ncclCommSplit(world, rank % pc, rank / pc, &col, NULL));     /* rank in `col` = process-row index */
  gpn_dist_comm* comm = gpn_rccl_comm_create(row, col, world);
  if (!comm) return 6;
  if (!grid) comm->flags |= GPN_DIST_FORCE_COLLECTIVES;

  const int64_t bytes = gpn_dist_work_bytes(rank, pr, pc, n, d, dy, tile);
  if (bytes < 0) { fprintf(stderr, "bad grid / tile\n"); return 7; }
  double *X, *Y, *th, *work, *out4;
  CHECK(hipMalloc((void**)&X, sizeof(double) * n * d));
  CHECK(hipMalloc((void**)&Y, sizeof(double) * n));
  CHECK(hipMalloc((void**)&th, sizeof(theta)));
  CHECK(hipMalloc((void**)&work, (size_t)bytes));
  CHECK(hipMalloc((void**)&out4, 4 * sizeof(double)));
  CHECK(hipMemcpy(X, x, sizeof(double) * n * d, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Y, y, sizeof(double) * n, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(th, theta, sizeof(theta), hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  double host4[4];
  for (int it = 0; it < 2; ++it) {          /* the workspace is reusable: the call clears what it needs */
    GPN(gpn_dist_lml_forward(s, comm, rank, pr, pc, GPN_RBF, X, n, d, Y, dy, th, th + 1, 1, th + 2, tile, work, bytes, out4));
    CHECK(hipMemcpyAsync(host4, out4, sizeof(host4), hipMemcpyDeviceToHost, s));
    CHECK(hipStreamSynchronize(s));
  }
  if (rank == 0) printf("lml=%.12f info=%.0f workspace_mb=%.1f\n", host4[2], host4[3], bytes / 1048576.0);
  GPN(gpn_release_stream(s));
  CHECK(hipStreamDestroy(s));
  gpn_rccl_comm_destroy(comm);
  ncclCommDestroy(row); ncclCommDestroy(col); ncclCommDestroy(world);
  return 0;
}
