/* capi_slab_demo.c -- a slab-decomposed run driven from plain C through
 * include/sf_hip.h alone (no Python, no torch, no MPI): the parent forks one
 * process per rank BEFORE anything touches the GPU; every rank builds its plan
 * (option "slab=lo:hi:halo"), registers its slab buffers with the library's
 * peer-to-peer transport (sf_halo_*), trades the 256-byte buffer descriptions with
 * its neighbours over socket pairs, and executes the chain: per launch, push the
 * boundary planes into the neighbours' ghost planes, compute the interior beside
 * the transfer, then the boundary planes.  Results go to <out>.<rank> (raw planes
 * the rank owns); tests/test_capi_slab.py compares them with the oracle.
 *
 * The reference's counterpart is `mpirun -n N bin/run_distributed_program.py`
 * (bin/run_distributed_program.py:98-100,283-299).  All ranks use device 0 here
 * (a one-GPU test box); on a node with N GPUs pass the rank as the device.
 *
 *   cc -I include tests/capi_slab_demo.c -L stencilflow_amd/csrc -lsf_hip -o demo
 *   ./demo <sfir file> <input.dat> <out prefix> <world> <n0> <plane_bytes> <halo> [mode]
 * mode "deep": the per-launch loop below is replaced by ONE call of the library's own
 * deep-halo schedule, sf_plan_execute_decomposed (halo = several launches' reach).
 * mode "rccl" / "rccl-deep": the same two forms over the library's RCCL rung
 * (sf_halo_rccl_id on rank 0, the id relayed rank to rank over the sockets,
 * sf_halo_use_rccl on every rank: grouped ncclSend / ncclRecv instead of DMA pushes) --
 * one GPU per rank (device = rank).  mode "rccl-self" (a one-GPU box): only rank 1 of
 * <world> runs, on a communicator of its own -- every halo it sends comes back to it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include "sf_hip.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    if ((call) < 0) {                                                      \
      fprintf(stderr, "rank %d: %s: %s\n", rank, #call, sf_last_error()); \
      return 2;                                                            \
    }                                                                      \
  } while (0)

static char* read_file(const char* path, size_t* size) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)n + 1);
  if (fread(buf, 1, (size_t)n, f) != (size_t)n) {
    fclose(f);
    free(buf);
    return NULL;
  }
  buf[n] = 0;
  fclose(f);
  if (size) *size = (size_t)n;
  return buf;
}

/* exchange one blob with a neighbour over its socket (both sides send, then receive) */
static int trade(int fd, const char* mine, char* theirs) {
  if (write(fd, mine, SF_HALO_BLOB_BYTES) != SF_HALO_BLOB_BYTES) return -1;
  size_t got = 0;
  while (got < SF_HALO_BLOB_BYTES) {
    ssize_t n = read(fd, theirs + got, SF_HALO_BLOB_BYTES - got);
    if (n <= 0) return -1;
    got += (size_t)n;
  }
  return 0;
}

/* a fixed number of bytes to / from a neighbour */
static int send_all(int fd, const char* p, size_t n) { return write(fd, p, n) == (ssize_t)n ? 0 : -1; }
static int recv_all(int fd, char* p, size_t n) {
  size_t got = 0;
  while (got < n) {
    ssize_t k = read(fd, p + got, n - got);
    if (k <= 0) return -1;
    got += (size_t)k;
  }
  return 0;
}

static int run_rank(int rank, int world, int down_fd, int up_fd, const char* sfir, const char* input,
                    size_t input_bytes, const char* out_prefix, int n0, size_t plane_bytes, int halo,
                    const char* session, int deep, int rccl /* 0 no, 1 one GPU per rank, 2 self-loop */) {
  const int lo = (int)((long long)n0 * rank / world), hi = (int)((long long)n0 * (rank + 1) / world);
  char opt[96]; /* (fourth field: the extent every rank plans with -- the thinnest slab) */
  snprintf(opt, sizeof opt, "slab=%d:%d:%d:%d", lo, hi, halo, n0 / world);
  const int device = rccl == 1 ? rank : 0; /* RCCL wants a GPU per rank; the other modes share device 0 */
  sf_plan* plan = NULL;
  CHECK(sf_plan_create(sfir, device, opt, &plan));
  sf_halo* link = NULL;
  CHECK(sf_halo_create(rank, world, session, device, 20000, &link));
  if (rccl) {
    char id[SF_HALO_RCCL_ID_BYTES];
    if (rccl == 2 || rank == 0) CHECK(sf_halo_rccl_id(id));
    if (rccl == 1) { /* the id travels up the chain of ranks */
      if (rank > 0 && recv_all(down_fd, id, sizeof id) != 0) return 3;
      if (rank < world - 1 && send_all(up_fd, id, sizeof id) != 0) return 3;
    }
    CHECK(sf_halo_use_rccl(link, id, rccl == 2 ? 0 : rank, rccl == 2 ? 1 : world));
  }
  for (int b = 0; b < sf_plan_num_buffers(plan); ++b) {
    void* base;
    size_t pb;
    int planes;
    CHECK(sf_plan_buffer_info(plan, b, &base, &pb, &planes));
    if (planes == 1) continue; /* not split into slabs */
    char mine[SF_HALO_BLOB_BYTES], lower[SF_HALO_BLOB_BYTES], upper[SF_HALO_BLOB_BYTES];
    CHECK(sf_halo_export(link, b, base, pb, hi - lo, halo, mine));
    if (rccl != 2) {
      if (rank > 0 && trade(down_fd, mine, lower) != 0) return 3;
      if (rank < world - 1 && trade(up_fd, mine, upper) != 0) return 3;
    }
    CHECK(sf_halo_connect(link, b, rank > 0 && rccl != 2 ? lower : NULL, rank < world - 1 && rccl != 2 ? upper : NULL));
  }
  if (input_bytes != (size_t)n0 * plane_bytes) {
    fprintf(stderr, "input file has %zu bytes, expected %zu\n", input_bytes, (size_t)n0 * plane_bytes);
    return 4;
  }
  const void* ins[1] = {input + (size_t)lo * plane_bytes};
  CHECK(sf_plan_upload(plan, ins));
  void* cs = NULL;
  CHECK(sf_plan_stream(plan, &cs));
  const int steps = deep ? 0 : sf_plan_num_steps(plan);
  if (deep) CHECK(sf_plan_execute_decomposed(plan, link, 1));
  for (int s = 0; s < steps; ++s) {
    int buf = -1, depth = 0;
    CHECK(sf_plan_step_halo(plan, s, &buf, &depth));
    if (depth > 0) CHECK(sf_halo_start(link, buf, depth, cs)); /* boundary planes -> the neighbours' ghosts */
    CHECK(sf_plan_execute_step(plan, s, 3, NULL));              /* interior, beside the transfer */
    if (depth > 0) CHECK(sf_halo_finish(link, buf, cs));
    CHECK(sf_plan_execute_step(plan, s, 1, NULL));              /* planes next to the lower boundary */
    CHECK(sf_plan_execute_step(plan, s, 2, NULL));              /* ... and the upper one */
  }
  if (rccl) CHECK(sf_halo_check(link)); /* RCCL: the bounded wait for the exchanges comes before any wait for the device */
  CHECK(sf_plan_synchronize(plan));
  CHECK(sf_halo_check(link));
  size_t out_bytes = sf_plan_output_bytes(plan, 0);
  char* out = (char*)malloc(out_bytes);
  void* outs[1] = {out};
  CHECK(sf_plan_download(plan, outs));
  char path[512];
  snprintf(path, sizeof path, "%s.%d", out_prefix, rank);
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(out, 1, out_bytes, f) != out_bytes) return 5;
  fclose(f);
  free(out);
  /* the neighbours may still be reading this rank's flag page: leave together */
  char token = 1, other;
  if (rccl == 2) down_fd = up_fd = -1;
  if (down_fd >= 0 && rank > 0 && (write(down_fd, &token, 1) != 1 || read(down_fd, &other, 1) != 1)) return 6;
  if (up_fd >= 0 && rank < world - 1 && (write(up_fd, &token, 1) != 1 || read(up_fd, &other, 1) != 1)) return 6;
  CHECK(sf_halo_destroy(link));
  CHECK(sf_plan_destroy(plan));
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 8 && argc != 9) {
    fprintf(stderr, "usage: %s <sfir> <input.dat> <out prefix> <world> <n0> <plane_bytes> <halo> "
                    "[deep|rccl|rccl-deep|rccl-self]\n", argv[0]);
    return 1;
  }
  const char* mode = argc == 9 ? argv[8] : "";
  const int deep = strcmp(mode, "deep") == 0 || strcmp(mode, "rccl-deep") == 0;
  const int rccl = strcmp(mode, "rccl-self") == 0 ? 2 : strncmp(mode, "rccl", 4) == 0 ? 1 : 0;
  const int world = atoi(argv[4]), n0 = atoi(argv[5]), halo = atoi(argv[7]);
  const size_t plane_bytes = (size_t)atoll(argv[6]);
  size_t input_bytes = 0;
  char* sfir = read_file(argv[1], NULL);
  char* input = read_file(argv[2], &input_bytes);
  if (!sfir || !input || world < 1 || world > 8) {
    fprintf(stderr, "cannot read the program or the input\n");
    return 1;
  }
  /* links[r] connects rank r (its "up" side) with rank r + 1 (its "down" side) */
  int links[8][2];
  for (int r = 0; r + 1 < world; ++r)
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, links[r]) != 0) return 1;
  char session[64];
  snprintf(session, sizeof session, "capi%d", (int)getpid());
  pid_t pids[8];
  const int first = rccl == 2 ? 1 : 0, last = rccl == 2 ? 2 : world; /* self-loop: rank 1 alone */
  if (rccl == 2 && world < 3) {
    fprintf(stderr, "rccl-self plays rank 1 of at least 3\n");
    return 1;
  }
  for (int r = first; r < last; ++r) {
    pids[r] = fork(); /* nothing has initialised the GPU yet */
    if (pids[r] == 0) {
      const int down = r > 0 ? links[r - 1][1] : -1, up = r < world - 1 ? links[r][0] : -1;
      const int rc = run_rank(r, world, down, up, sfir, input, input_bytes, argv[3], n0, plane_bytes, halo, session, deep, rccl);
      _exit(rc);
    }
  }
  int failed = 0;
  for (int r = first; r < last; ++r) {
    int status = 0;
    waitpid(pids[r], &status, 0);
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
      fprintf(stderr, "rank %d failed (status %d)\n", r, status);
      failed = 1;
    }
  }
  return failed;
}
