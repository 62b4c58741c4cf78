/*
 * aslp_parallel.h -- C handle API of libaslp_parallel.so: the model-sync layer (boundary B6 of SURVEY.md §8b) for
 * hosts that are not C++ (bench.py, tests).  C++ callers link kaldi-aslp_amd/parallel/{comm,workers}.h directly.
 *
 *   aslp_comm_*    what MpiNode gives the reference's workers (src/aslp-parallel/mpi-node.h:19-97): rank, size, barrier,
 *                  sum all-reduce, point-to-point transfers -- on RCCL over xGMI with the buffers staying in HBM, one
 *                  process per GPU, ranks meeting through a rendezvous file instead of mpirun.
 *   aslp_worker_*  the workers of src/aslp-parallel/itf.h:26-42: InitParam / Synchronize / Stop for
 *                  "bsp" (bsp-worker.cc:33-65), "bmuf" (bmuf-worker.cc:37-68), "easgd" (easgd-worker.cc:37-80),
 *                  "asgd" (asgd-worker.cc:37-71);  aslp_server_* the rank-0 side of the served protocols
 *                  (easgd-server.cc:37-86, asgd-server.cc:39-102, masgd-server.cc:39-118).
 *
 * Every function returns 0 on success and non-zero on error (the reference throws KALDI_ERR); the message is in
 * aslp_parallel_last_error().  Device pointers are fp32 (fp64 where named) in the caller's HBM; the workers alias the
 * model's parameter tensors (Nnet::GetGpuParams) and never own them.  Launches go to the calling thread's current
 * stream (aslp_set_stream of libaslp_hip.so).
 */
#ifndef ASLP_PARALLEL_H_
#define ASLP_PARALLEL_H_
#include <stddef.h>
#include <stdint.h>

#include "aslp_nnet.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct aslp_comm_s *aslp_comm_t;
typedef struct aslp_worker_s *aslp_worker_t;

const char *aslp_parallel_last_error(void);

/* rank / num_nodes < 0: taken from the launcher's environment (OMPI_COMM_WORLD_*, PMI_*, RANK / WORLD_SIZE).
 * id_file: rendezvous file on a filesystem every rank sees (may be NULL for a group of one).  token (may be NULL): ties
 * the file to this launch; default ASLP_COMM_TOKEN / TORCHELASTIC_RUN_ID / SLURM_JOB_ID.  timeout_s bounds both the wait
 * for the file and ncclCommInitRank.  Create the communicator BEFORE the model is allocated. */
int aslp_comm_create_rccl(int rank, int num_nodes, const char *id_file, const char *token, int timeout_s, aslp_comm_t *out);
/* The same communicator for ranks that are separate processes on ANY devices -- all on one GPU included, which RCCL refuses: the same
 * rendezvous file and control pipe, tensors staged through a POSIX shared-memory segment (kaldi-aslp_amd/parallel/comm.cpp ShmComm).
 * A functional transport for single-GPU boxes; the worker tools take it with --comm-transport=shm / ASLP_COMM_TRANSPORT=shm. */
int aslp_comm_create_shm(int rank, int num_nodes, const char *id_file, const char *token, int timeout_s, aslp_comm_t *out);
void aslp_comm_free(aslp_comm_t c);
int aslp_comm_rank(aslp_comm_t c);        /* MpiNode::Rank      mpi-node.h:40 */
/* what the transport itself reports: the size of the live group (RCCL: ncclCommCount) and "rccl" | "shm" | "threads" -- so that a bench
 * line can prove which transport carried a run and that every rank joined it (no reference counterpart) */
int aslp_comm_ranks_seen(aslp_comm_t c);
const char *aslp_comm_transport(aslp_comm_t c);
int aslp_comm_num_nodes(aslp_comm_t c);   /* MpiNode::NumNodes  mpi-node.h:43 */
int aslp_comm_barrier(aslp_comm_t c);     /* MpiNode::Barrier   mpi-node.h:46 */
/* MpiNode::AllReduce (mpi-node.h:52-75) without the host round trip: in place on device memory */
int aslp_comm_allreduce_sum_f32(aslp_comm_t c, float *dev, size_t n);
int aslp_comm_allreduce_sum_f64(aslp_comm_t c, double *dev, size_t n);
/* small host-side values (the sample counts of bsp-worker.cc:34-36) */
int aslp_comm_allreduce_sum_host_i32(aslp_comm_t c, int32_t *host, size_t n);
int aslp_comm_allreduce_sum_host_f64(aslp_comm_t c, double *host, size_t n);
/* MPI_Send / MPI_Recv of easgd-worker.cc:44-58 as ncclSend / ncclRecv; exchange = both directions in one group */
int aslp_comm_send_f32(aslp_comm_t c, int peer, float *dev, size_t n);
int aslp_comm_recv_f32(aslp_comm_t c, int peer, float *dev, size_t n);
int aslp_comm_exchange_f32(aslp_comm_t c, int peer, float *send_dev, float *recv_dev, size_t n);

/* kind: "bsp" | "bmuf" (p0 = learn rate, p1 = block momentum) | "easgd" (p0 = alpha) | "asgd" */
int aslp_worker_create(aslp_comm_t c, const char *kind, float p0, float p1, aslp_worker_t *out);
void aslp_worker_free(aslp_worker_t w);
int aslp_worker_init_param(aslp_worker_t w, float *const *dev_ptrs, const int *sizes, int n);   /* ISynchronizer::InitParam itf.h:30 */
int aslp_worker_init_param_nnet(aslp_worker_t w, aslp_nnet_t net);                              /* = InitParam(nnet.GetGpuParams()) */
/* ISynchronizer::Synchronize (itf.h:33): *more = 0 once every worker has run out of data */
int aslp_worker_synchronize(aslp_worker_t w, int num_worker_samples, int *more);
int aslp_worker_stop(aslp_worker_t w);                                                          /* itf.h:35 */
/* rank 0 of a served protocol: kind "easgd" (p0 = alpha) | "asgd" (p0 = alpha, sync_period) | "masgd" (p0 = alpha, p1 = momentum);
 * serves until every worker has sent "finished" (easgd-server.cc:63-86) */
int aslp_server_run(aslp_comm_t c, const char *kind, float p0, float p1, int sync_period, float *const *dev_ptrs, const int *sizes, int n);

#ifdef __cplusplus
}
#endif
#endif
