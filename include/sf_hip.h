/* sf_hip.h — C ABI of libsf_hip.so, the MI355X (gfx950) execution backend for
 * StencilFlow stencil-chain programs.
 *
 * Boundary being replaced (paths relative to the reference tree):
 *   The reference's driver obtains a compiled program from DaCe and calls it
 *   through ctypes:  `program = sdfg.compile()`            stencilflow/run_program.py:120-123
 *                    `program(**dace_args)` x repetitions   stencilflow/run_program.py:164-178
 *   i.e. an init / call / exit triple over caller-owned, C-contiguous NumPy
 *   buffers (arrays keyed `<name>_host`, 0-D inputs passed by value).
 *   `sf_plan_create / sf_plan_run / sf_plan_destroy` are that triple; the
 *   record DaCe receives (`generate_sdfg`, stencilflow/sdfg_generator.py:219-577,
 *   per operator `_generate_stencil`, :68-176) is passed here as SFIR text
 *   (grammar: stencilflow_amd/lowering.py).
 *
 * Conventions: plain C types only; every function returns 0 on success and a
 * negative sf_status on failure, with a thread-local message available from
 * sf_last_error(); nothing throws across the boundary.  Host buffers belong
 * to the caller (inputs are only read, outputs only written, during the call);
 * all device memory, streams and code objects belong to the plan.  A plan is
 * not re-entrant; distinct plans may be used from distinct threads.
 */
#ifndef SF_HIP_H
#define SF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sf_plan sf_plan;

enum sf_status {
  SF_OK = 0,
  SF_ERR_INVALID = -1,     /* bad argument / malformed SFIR  (Python: ValueError)   */
  SF_ERR_UNSUPPORTED = -2, /* valid program the backend cannot run (ValueError)    */
  SF_ERR_COMPILE = -3,     /* hipRTC rejected generated code       (RuntimeError)  */
  SF_ERR_DEVICE = -4,      /* HIP runtime error / no GPU           (RuntimeError)  */
  SF_ERR_STATE = -5        /* call order violated                  (RuntimeError)  */
};

/* ABI version of this header (major*1000 + minor). */
int sf_version(void);

/* Message of the last failure on the calling thread ("" if none). */
const char* sf_last_error(void);

/* Number of visible HIP devices, or a negative status. */
int sf_device_count(void);

/* Build a plan from SFIR text: analysis, kernel generation and hipRTC
 * compilation for gfx950.  No GPU is touched until the first run/upload, so
 * this succeeds on a machine without a device (used by the build check).
 * `options` is a ';'-separated list of key=value tuning overrides or NULL
 * (keys: see DESIGN.md §Options).                     replaces: sdfg.compile(),
 * stencilflow/run_program.py:118-128 */
int sf_plan_create(const char* sfir_text, int device, const char* options,
                   sf_plan** out_plan);

int sf_plan_destroy(sf_plan* plan);

/* Code objects are cached in the process and on disk ($SF_HIP_CACHE_DIR, default
 * ~/.cache/stencilflow_amd; "off" disables; the role of the reference's
 * `-use-cached-sdfg`, stencilflow/run_program.py:69-73,83-88).  Counters of this
 * process since it loaded the library: objects taken from disk, objects
 * compiled, and cached objects that were rejected by the loader and rebuilt.
 * `drop_process_level` != 0 also empties the in-process level, so that the next
 * plan goes to disk again (tests).  Any pointer may be NULL. */
int sf_code_cache_stats(long* disk_hits, long* compiled, long* rebuilt,
                        int drop_process_level);

/* Plan-time self-check.  Before a plan's first use every FUSED kernel whose code object
 * carries no verdict yet is run once on seeded data (planes next to both ends of the
 * slab) and compared bit for bit with the same operators evaluated one by one by the
 * plain generic kernel; the verdict is stored with the code object in both cache levels.
 * A kernel that differs is never launched: the first use fails with SF_ERR_UNSUPPORTED and
 * the planner skips the object from then on.  $SF_HIP_SELF_CHECK=0 disables the check.
 * sf_self_checks_run: fused kernels checked by this process so far;
 * sf_plan_kernel_verdict: 0 not checked (yet), 1 passed, 2 failed. */
long sf_self_checks_run(void);
int sf_plan_kernel_verdict(const sf_plan* plan, int index);

/* Introspection: the argument lists `sf_plan_run` expects, in order.
 * Arrays first (program order), then nothing else: 0-D inputs are scalars. */
int sf_plan_num_inputs(const sf_plan* plan);   /* array inputs   */
int sf_plan_num_scalars(const sf_plan* plan);  /* 0-D inputs     */
int sf_plan_num_outputs(const sf_plan* plan);
const char* sf_plan_input_name(const sf_plan* plan, int index);
const char* sf_plan_scalar_name(const sf_plan* plan, int index);
const char* sf_plan_output_name(const sf_plan* plan, int index);
/* bytes of input/output array `index` (full array, C order) */
size_t sf_plan_input_bytes(const sf_plan* plan, int index);
size_t sf_plan_output_bytes(const sf_plan* plan, int index);

/* Run-time values of the 0-D inputs, as doubles (converted to each scalar's
 * declared type).  Must be set before the first run if the program has any. */
int sf_plan_set_scalars(sf_plan* plan, const double* values, int count);

/* The drop-in call: copy inputs host->device, execute the whole chain
 * `repetitions` times, copy outputs device->host, synchronously.
 *                                replaces: program(**dace_args),
 *                                stencilflow/run_program.py:170-178 */
int sf_plan_run(sf_plan* plan, const void* const* host_inputs,
                void* const* host_outputs, int repetitions);

/* Device-resident variant used for measurement and by the multi-GPU driver:
 * inputs stay in HBM between executions. */
int sf_plan_upload(sf_plan* plan, const void* const* host_inputs);
int sf_plan_execute(sf_plan* plan, int repetitions); /* asynchronous */
int sf_plan_synchronize(sf_plan* plan);
int sf_plan_download(sf_plan* plan, void* const* host_outputs);

/* HIP-event time of the last sf_plan_execute (whole chain, all repetitions),
 * in milliseconds; valid after sf_plan_synchronize. */
int sf_plan_elapsed_ms(sf_plan* plan, double* ms);

/* Launch schedule of one execution of the chain. */
int sf_plan_num_launches(const sf_plan* plan);
/* Distinct generated kernels. */
int sf_plan_num_kernels(const sf_plan* plan);
/* Name of generated kernel `index` as it appears in rocprofv3 traces. */
const char* sf_plan_kernel_name(const sf_plan* plan, int index);
/* Generated HIP source of kernel `index` (for inspection / offline hipcc). */
const char* sf_plan_kernel_source(const sf_plan* plan, int index);
/* Per-kernel timing of the last execution when the option "profile=1" is set:
 * number of launches of kernel `index`, their summed HIP-event duration (ms),
 * cell updates and algorithmic bytes (2*sizeof(dtype) per update) per launch. */
int sf_plan_kernel_stats(sf_plan* plan, int index, int* launches,
                         double* total_ms, double* updates_per_launch,
                         double* algorithmic_bytes_per_launch);
/* Turn per-launch HIP-event profiling on / off after plan creation (same effect as the
 * option "profile=1"): bench.py times one extra, untimed chain execution this way so that
 * the timed region carries no events.  Synchronises the plan's stream first. */
int sf_plan_set_profile(sf_plan* plan, int on);
/* The device compiler of this process: hipRTC and runtime versions and the libamd_comgr file
 * the process compiles through (with its version).  hipRTC binds comgr by soname, so the copy
 * loaded first does the work; $SF_HIP_COMGR=<file> pins it: plan creation fails
 * (SF_ERR_STATE) in a process whose comgr is another file.  The same text ends
 * sf_plan_describe and is part of the code cache's key.  (The reference compiles each
 * program with one toolchain, stencilflow/run_program.py:118-128.) */
const char* sf_compiler_id(void);
/* Index of the compiled kernel (sf_plan_kernel_name ...) that launch `step` runs. */
int sf_plan_step_kernel(const sf_plan* plan, int step);
/* The code object of kernel `index` as the plan will load it (an ELF for gfx950: what hipRTC
 * compiled, or the cached copy) and the extra compiler flags it was built with; the
 * pointers stay valid for the life of the plan.  tools/isa_stats.py disassembles it. */
int sf_plan_kernel_object(const sf_plan* plan, int index, const void** data, size_t* bytes,
                          const char** flags);
/* With profiling on: shortest, median and longest HIP-event duration (ms) of the profiled
 * launches of kernel `index` since the counters were last reset (the first 16384 are kept).
 * SF_ERR_STATE when none was profiled.  bench.py reports them as roofline.min_us /
 * median_us / max_us (a median, not only a mean). */
int sf_plan_kernel_launch_times(sf_plan* plan, int index, double* min_ms,
                                double* median_ms, double* max_ms);
/* With profiling on: planes of the outermost dimension written by the launches of kernel
 * `index` in the executions since the counters were last reset (sf_plan_execute,
 * sf_plan_set_profile) -- a decomposed run launches over plane ranges of different
 * lengths, so bytes per launch follow from this, not from the launch count. */
int sf_plan_kernel_planes(sf_plan* plan, int index, double* planes);
/* Register / LDS footprint of generated kernel `index`, read from the compiled
 * code object's metadata (-1 where the compiler did not report a field). */
int sf_plan_kernel_resources(const sf_plan* plan, int index, int* vgprs,
                             int* agprs, int* vgpr_spills, int* scratch_bytes,
                             int* lds_bytes);
/* The plan options this library accepts, one `key=<value>  meaning` line each (static text).  sf_plan_create refuses
 * a key that is not listed (SF_ERR_INVALID) -- the reference's run_program has no such knobs (run_program.py:19-34);
 * they pin what the planner would choose (tile shapes, fusion depth, kernel families) for tests and measurements. */
const char* sf_describe_options(void);
/* Human-readable description of the schedule (groups, tiles, buffers). */
const char* sf_plan_describe(const sf_plan* plan);

/* ---- slab decomposition (multi-GPU; one plan per rank) --------------------
 * The outermost dimension is split into contiguous slabs.  A rank's plan is
 * created with the option "slab=<lo>:<hi>:<halo>[:<extent>]" and works on local
 * arrays of (hi-lo+2*halo) planes.  All ranks must build the same launches (they
 * exchange the same planes): with slabs of unequal height pass the same <extent>
 * -- e.g. the thinnest slab's height -- on every rank; the planner then derives
 * everything that depends on the slab's height from it.  One chain execution is a sequence of steps; before
 * step s the planes listed by sf_plan_step_halo must have been exchanged with
 * the neighbouring ranks for the buffer it names. */
int sf_plan_num_steps(const sf_plan* plan);
/* depth (planes) and device buffer id of the field step `step` reads across
 * slab boundaries; depth 0 if the step needs no exchange. */
int sf_plan_step_halo(const sf_plan* plan, int step, int* buffer_id,
                      int* depth);
/* Device buffers step `step` reads (ids written to buffer_ids[0..capacity),
 * return value = their number) and the buffer it writes. */
int sf_plan_step_inputs(const sf_plan* plan, int step, int* buffer_ids,
                        int capacity);
int sf_plan_step_output(const sf_plan* plan, int step);
/* All buffers step `step` writes (same convention as sf_plan_step_inputs).  One for every launch but a DAG group
 * (round 4: a fork's branches, a join, an intermediate with several readers fused into one launch), which
 * materialises every field something outside the group reads; sf_plan_step_output is the first of them. */
int sf_plan_step_outputs(const sf_plan* plan, int step, int* buffer_ids, int capacity);
/* Execute one step on `stream` (a hipStream_t, or NULL for the plan's own).
 * part: 0 = whole slab, 1 = planes adjacent to the lower slab boundary,
 *       2 = planes adjacent to the upper boundary, 3 = interior only. */
int sf_plan_execute_step(sf_plan* plan, int step, int part, void* stream);
/* Execute one step over explicit plane ranges [i_begin, i_end) and (optionally,
 * i_begin2 < i_end2) [i_begin2, i_end2), in owned-plane coordinates: 0 is the
 * first owned plane, negative values and values >= the slab height address
 * halo planes (the launch then recomputes planes a neighbour also owns, which
 * is how a deep halo is made to last several launches). */
int sf_plan_execute_step_ranges(sf_plan* plan, int step, int i_begin, int i_end,
                                int i_begin2, int i_end2, void* stream);
/* Keep `cus` of the 256 compute units free of this plan's blocks in the launches
 * that follow (0 = use them all): the chunking of the star kernels then fills
 * 256 - cus units.  The slab runner sets it around the interior launch that
 * runs beside a halo exchange, so that the RCCL copy kernels find free units
 * instead of queueing behind 200-microsecond blocks that leave no registers. */
int sf_plan_set_reserved_cus(sf_plan* plan, int cus);
/* The HIP stream (hipStream_t) the plan launches on when a call is given NULL. */
int sf_plan_stream(sf_plan* plan, void** stream);
/* Number of device buffers of the plan (ids 0 .. n-1). */
int sf_plan_num_buffers(const sf_plan* plan);
/* Device address, plane size in bytes and plane count of device buffer `id`. */
int sf_plan_buffer_info(const sf_plan* plan, int buffer_id, void** device_ptr,
                        size_t* plane_bytes, int* planes);
/* Buffer ids holding input `index` / output `index` at start / end of a chain. */
int sf_plan_input_buffer(const sf_plan* plan, int index);
int sf_plan_output_buffer(const sf_plan* plan, int index);

/* ---- halo transport through host memory shared by the ranks of one node ----
 * The primitives of the spare transport of the slab decomposition (used when
 * RCCL cannot connect the ranks): a rank copies the planes it sends into a
 * pinned buffer that its neighbour has mapped too (POSIX shared memory), raises
 * a flag word in that buffer from its stream, and the neighbour's stream waits
 * for the flag before copying the planes into its ghost planes.  Everything is
 * stream-ordered on both sides; no host thread takes part after enqueueing.
 * (The reference moves data between devices with SMI streams and MPI barriers,
 * stencilflow/sdfg_generator.py:782-1000, bin/run_distributed_program.py:283-299.) */
/* Pin `bytes` of host memory at `ptr` (e.g. an mmap of /dev/shm) for device
 * access / undo it.  *device_ptr receives the address kernels must use for the
 * range (flag words); copies take either address. */
int sf_host_register(void* ptr, size_t bytes, void** device_ptr);
int sf_host_unregister(void* ptr);
/* dst <- src on `stream` (either side may be pinned host or device memory). */
int sf_copy_async(void* dst, const void* src, size_t bytes, void* stream);
/* Store `value` to the flag word (system scope) once everything enqueued on
 * `stream` before it has completed. */
int sf_flag_set(void* stream, unsigned int* flag, unsigned int value);
/* Hold `stream` until the flag word has reached `value` (counting comparison:
 * *flag - value >= 0 as a signed difference).  After `timeout_ms` the wait gives
 * up, stores 1 to *status (pinned host or device memory, may be NULL) and lets
 * the stream continue: a waiting kernel always terminates. */
int sf_flag_wait(void* stream, const unsigned int* flag, unsigned int value,
                 unsigned int timeout_ms, unsigned int* status);

/* ---- halo transport of a slab-decomposed run, owned by the library --------------
 * One rank's end of the neighbour exchange.  Two rungs behind one handle:
 *  - peer-to-peer pushes (default): a rank PUSHES the planes next to a slab boundary
 *    straight into its neighbour's ghost planes -- device memory of the neighbour's plan,
 *    mapped through a HIP IPC handle -- with DMA copies (no compute units; over xGMI
 *    between the GPUs of a node), ordered by flag words in a page of host memory the
 *    ranks of the node share;
 *  - RCCL (sf_halo_use_rccl): grouped ncclSend / ncclRecv of the same planes on the
 *    transport's stream (librccl is loaded on demand with dlopen).
 * Nothing but this header is needed to drive a decomposed run from C: create a plan per
 * rank (option "slab=..."), [sf_halo_use_rccl,] sf_halo_export its slab buffers, hand
 * the blobs to the neighbouring ranks by whatever means the caller has (MPI, a file,
 * torch.distributed ...), sf_halo_connect, then alternate sf_halo_start /
 * sf_plan_execute_step_ranges / sf_halo_finish -- or let sf_plan_execute_decomposed run
 * the whole schedule.
 * (Role in the reference: the SMI remote streams between devices,
 * stencilflow/sdfg_generator.py:848-891, and the MPI rank bookkeeping of
 * bin/run_distributed_program.py:98-100,283-299.) */
typedef struct sf_halo sf_halo;
#define SF_HALO_BLOB_BYTES 256
/* `session`: a name unique to this run, the same on all ranks (names the shared
 * flag pages).  timeout_ms: how long a stream waits for a neighbour before it
 * gives up and marks the transport failed (sf_halo_check); 0 = 20 s. */
int sf_halo_create(int rank, int world, const char* session, int device,
                   unsigned int timeout_ms, sf_halo** out);
int sf_halo_destroy(sf_halo* halo);
/* Register slab buffer `key` (0..59; e.g. the plan's buffer id): `device_base` is
 * the start of its allocation (sf_plan_buffer_info), holding n_local + 2*halo
 * planes of plane_bytes.  Writes SF_HALO_BLOB_BYTES to `blob`: the description the
 * neighbouring ranks need. */
int sf_halo_export(sf_halo* halo, int key, void* device_base, size_t plane_bytes,
                   int n_local, int halo_planes, void* blob);
/* Map the neighbours' buffers `key` from their blobs (NULL where there is no
 * neighbour).  Collective with the neighbours' calls only through the blobs. */
int sf_halo_connect(sf_halo* halo, int key, const void* lower_blob,
                    const void* upper_blob);
/* Start exchange of `depth` planes per direction of buffer `key`: waits (on the
 * transport's streams) for everything enqueued on `compute_stream` so far, then
 * pushes the owned planes next to each boundary into the neighbours' ghost planes
 * and waits for theirs.  Asynchronous; every rank calls it at the same point of
 * the schedule. */
int sf_halo_start(sf_halo* halo, int key, int depth, void* compute_stream);
/* Make `compute_stream` wait until the exchange started last on `key` is done. */
int sf_halo_finish(sf_halo* halo, int key, void* compute_stream);
/* Fails (SF_ERR_DEVICE) if a wait of this rank -- or a neighbour's wait for this rank --
 * has timed out (peer-to-peer; call after synchronising), or if RCCL reports an
 * asynchronous error.  RCCL rung: ncclSend / ncclRecv have no time limit, so this call
 * is also the bound -- it waits ON THE HOST, at most `timeout_ms`, until every exchange
 * started so far has arrived; call it BEFORE waiting for the device.  After a time-out
 * or an error the communicator has been ended (ncclCommAbort: its kernels leave the
 * streams, later synchronisations return) and the transport stays failed. */
int sf_halo_check(sf_halo* halo);
/* The caller gives the transport up (e.g. the communicator did not form on every rank):
 * what RCCL holds is ended with ncclCommAbort, and sf_halo_destroy will not wait for it
 * (ncclCommDestroy of a half-formed communicator may wait for the missing ranks). */
int sf_halo_fail(sf_halo* halo);
/* For a handle another thread is still inside (an ncclCommInitRank that never returned):
 * it can be neither used nor destroyed; this releases the one thing that outlives the
 * process -- the name of its flag page in /dev/shm -- without touching the rest. */
int sf_halo_abandon(sf_halo* halo);
/* The RCCL rung.  One rank (any) obtains an id -- SF_HALO_RCCL_ID_BYTES bytes, an
 * ncclUniqueId -- and hands it to every rank of the run; every rank then calls
 * sf_halo_use_rccl BEFORE its first sf_halo_export (collective: ncclCommInitRank over
 * comm_size ranks).  comm_size is the transport's world and comm_rank its rank; for tests
 * on a one-GPU machine comm_size may be 1 with comm_rank 0: the rank then is its own lower
 * and upper neighbour and every halo it sends comes back to it.  Export / connect are
 * called as for the peer-to-peer rung (the blobs then carry geometry only).
 * SF_ERR_UNSUPPORTED when librccl cannot be loaded ($SF_RCCL_LIBRARY names it). */
#define SF_HALO_RCCL_ID_BYTES 128
int sf_halo_rccl_id(void* id_out);
int sf_halo_use_rccl(sf_halo* halo, const void* id, int comm_rank, int comm_size);
/* "p2p" or "rccl". */
const char* sf_halo_transport(const sf_halo* halo);
/* Refinements of sf_plan_execute_decomposed (both 0 by default): `reserved_cus` compute
 * units the launch beside a transfer leaves free (for the copy kernels of the RCCL rung:
 * the star kernels' blocks hold their unit for ~200 us); `early_exchange` != 0 starts an
 * exchange one launch ahead -- the launch before the one that needs fresh halos computes
 * its planes next to the slab boundaries first -- so that a transfer slower than one
 * interior launch has two launches of cover. */
int sf_halo_configure(sf_halo* halo, int reserved_cus, int early_exchange);
/* Exchange profile (what the reference's distributed driver reports as communication time,
 * bin/run_distributed_program.py:283-341 -- there per SMI channel, here per halo exchange): with `on` != 0 every
 * sf_halo_start records two timing events on the transport's streams -- when the launches the exchange waits for are
 * done, and when its planes have arrived.  sf_halo_exchange_times waits for the transport's streams and returns the
 * number of exchanges since the profile was switched on and their mean / longest duration in milliseconds: to be set
 * beside the duration of the interior launch an exchange has to hide behind (sf_plan_kernel_launch_times). */
int sf_halo_set_profile(sf_halo* halo, int on);
int sf_halo_exchange_times(sf_halo* halo, int* count, double* mean_ms, double* max_ms);
/* One rank's execution of the whole chain, `repetitions` times, with the deep-halo
 * schedule: the plan was created with "slab=lo:hi:H" and every slab
 * buffer registered with `halo` under its buffer id.  After an exchange of H planes a
 * launch of reach d also recomputes the H - d ghost planes that are still
 * computable, so the following launches need no communication; the launch that does
 * is split -- its interior runs beside the transfer, both boundary regions follow in
 * one launch.  Programs that are not a pure chain exchange every slab buffer a launch
 * reads across planes, at that launch's reach.  Asynchronous on the plan's stream;
 * every rank makes the same call. */
int sf_plan_execute_decomposed(sf_plan* plan, sf_halo* halo, int repetitions);

#ifdef __cplusplus
}
#endif
#endif /* SF_HIP_H */
