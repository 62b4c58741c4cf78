// comm.h -- the communicator under the model-sync workers: what MpiNode (src/aslp-parallel/mpi-node.h:19-97) provides
// to them -- rank, size, barrier, sum all-reduce -- with the parameter buffers staying in HBM.
//   RcclComm    one process per GPU, RCCL over xGMI (ncclAllReduce on device pointers, no host staging).  No MPI in the
//               launch path: ranks find each other through a file that rank 0 writes the ncclUniqueId into.
//   ShmComm     one process per rank on ANY devices, all on one GPU included: the same rendezvous file and control pipe, tensors
//               staged through a POSIX shared-memory segment.  What runs the process-per-rank machinery (server + workers as separate
//               OS processes) on a one-GPU box, where RCCL refuses two ranks on one device.
//   ThreadComm  N ranks as threads of ONE process on one GPU -- the harness the worker arithmetic is tested with.
#pragma once
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "base.h"

namespace aslp {

class Comm {
 public:
  virtual ~Comm() {}
  virtual int Rank() const = 0;
  virtual int NumNodes() const = 0;
  // how many ranks the transport itself reports in the group (RcclComm: ncclCommCount of the live communicator) and its name: a bench
  // line or a log can then prove which transport carried a run and that every rank really joined it
  virtual int RanksSeen() const { return NumNodes(); }
  virtual const char *Transport() const = 0;
  virtual void Barrier() = 0;
  virtual void AllReduceSum(float *dev, size_t n) = 0;        // in place, device memory
  virtual void AllReduceSum(double *dev, size_t n) = 0;       // in place, device memory
  virtual void AllReduceSumHost(int32 *host, size_t n) = 0;   // small host-side counts
  virtual void AllReduceSumHost(double *host, size_t n) = 0;
  // several device buffers in one grouped operation
  virtual void AllReduceSumMany(const std::vector<std::pair<float *, int>> &bufs) {
    for (auto &b : bufs) AllReduceSum(b.first, (size_t)b.second);
  }
  // ---- what the server-based protocols need (EASGD / ASGD / MASGD: rank 0 serves the others in ARRIVAL order) ----
  // Control channel, the MPI_Recv(MPI_ANY_SOURCE) of easgd-server.cc:44: a worker posts a small message to rank 0, rank 0
  // takes messages in the order they arrived.  (RCCL has no any-source receive: the order is decided on the host -- a
  // named pipe next to the rendezvous file for RcclComm, a queue for ThreadComm -- the tensors still move GPU to GPU.)
  typedef std::vector<std::pair<float *, int>> Buffers;
  virtual void PostToServer(int32 msg) = 0;
  virtual void WaitFromWorker(int *src, int32 *msg) = 0;
  // point-to-point transfers of whole parameter sets (device memory), each one grouped operation; Exchange = both ways at once
  virtual void Send(int peer, const Buffers &bufs) = 0;
  virtual void Recv(int peer, const Buffers &bufs) = 0;
  virtual void Exchange(int peer, const Buffers &send, const Buffers &recv) = 0;
};

// rank / world size of this process: --rank / --num-workers style explicit values win (>= 0), then the launcher's
// environment (OMPI_COMM_WORLD_*, PMI_*, RANK / WORLD_SIZE), else a single rank
void RankFromEnvironment(int *rank, int *num_nodes);

// id_file: rendezvous file; rank 0 creates it (atomically), the others wait for it (up to timeout_s, which also bounds
// ncclCommInitRank).  token: ties the file to one launch (default: ASLP_COMM_TOKEN / the launcher's job id); a file with
// another token, another world size or an age beyond ASLP_COMM_MAX_AGE_S is a leftover and is not joined.
Comm *NewRcclComm(int rank, int num_nodes, const std::string &id_file, int timeout_s = 900, const std::string &token = std::string());

// the same contract on a shared-memory segment (ranks may share a GPU); slot size per rank: ASLP_SHM_SLOT_MB (default 16)
Comm *NewShmComm(int rank, int num_nodes, const std::string &id_file, int timeout_s = 900, const std::string &token = std::string());
// the process-per-rank communicator the tools create: transport "rccl" (default) | "shm"; empty = ASLP_COMM_TRANSPORT, else rccl
Comm *NewProcessComm(const std::string &transport, int rank, int num_nodes, const std::string &id_file, int timeout_s = 900,
                     const std::string &token = std::string());

class ThreadCommGroup;
// one group, then one Comm per thread
std::shared_ptr<ThreadCommGroup> NewThreadCommGroup(int num_nodes);
Comm *NewThreadComm(std::shared_ptr<ThreadCommGroup> group, int rank);

}  // namespace aslp
