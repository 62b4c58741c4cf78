/* aslp_compat_kaldi_parallel.h -- the model-sync classes of src/aslp-parallel/ for callers written against the reference (seam B6 as a
 * source-level drop-in; included through the forwarding headers include/kaldi_compat/aslp-parallel/{itf,bsp-worker,...}.h).
 *
 * The reference's workers and servers find their peers through MPI_COMM_WORLD: `new BspWorker()`, `new EasgdWorker(alpha)`,
 * `new EasgdServer(alpha)` (itf.h:26-50 and the *-worker.h / *-server.h headers).  The engine's classes of the same names and methods
 * (kaldi-aslp_amd/parallel/workers.h) take the communicator as their first constructor argument.  What stands in for MPI_COMM_WORLD here is
 * ONE communicator per process, made on first use from what the launcher left in the environment -- rank and world size from
 * OMPI_COMM_WORLD_* / PMI_* / RANK + WORLD_SIZE (mpirun, srun and torchrun all set one of them), the rendezvous file from ASLP_COMM_FILE, the
 * transport from ASLP_COMM_TRANSPORT (rccl by default, one GPU per rank; shm lets ranks share a GPU) -- and the classes below are the engine's
 * with the reference's constructor signatures.  Link with -laslp_parallel. */
#ifndef ASLP_COMPAT_KALDI_PARALLEL_H_
#define ASLP_COMPAT_KALDI_PARALLEL_H_

#include <cstdlib>
#include <memory>
#include <string>

#include "aslp_compat_kaldi.h"
#include "workers.h"

namespace kaldi {
namespace aslp_nnet {

inline ::aslp::Comm *WorldComm() {   /* (created once, lives as long as the process: what MPI_Init .. MPI_Finalize bracket in the reference) */
  static std::unique_ptr< ::aslp::Comm> world = [] {
    int rank = -1, n = -1;
    ::aslp::RankFromEnvironment(&rank, &n);
    const char *file = getenv("ASLP_COMM_FILE");
    if (n > 1 && (file == nullptr || !*file))
      ASLP_ERR << "more than one rank: ASLP_COMM_FILE must name the rendezvous file of this launch (a path every rank can reach)";
    return std::unique_ptr< ::aslp::Comm>(::aslp::NewProcessComm("", rank, n, file ? file : ""));
  }();
  return world.get();
}

/* a served protocol needs its server: rank 0 runs aslp-nnet-train-server, the workers are ranks 1 .. N-1 (under MPI the reference's worker would
 * wait for a peer that does not exist) */
inline ::aslp::Comm *ServedWorkerComm(const char *type) {
  ::aslp::Comm *c = WorldComm();
  if (c->NumNodes() < 2 || c->Rank() == 0)
    ASLP_ERR << "worker type " << type << " needs aslp-nnet-train-server as rank 0 and the workers as ranks 1 .. N-1 (this is rank " << c->Rank() << " of "
             << c->NumNodes() << ")";
  return c;
}

using ::aslp::IWorker;
using ::aslp::IServer;
using ::aslp::OptimizerOption;
class BspWorker : public ::aslp::BspWorker { public: BspWorker() : ::aslp::BspWorker(WorldComm()) {} };
class BmufWorker : public ::aslp::BmufWorker { public: BmufWorker(float learn_rate, float momentum) : ::aslp::BmufWorker(WorldComm(), learn_rate, momentum) {} };
class SodWorker : public ::aslp::SodWorker { public: explicit SodWorker(const OptimizerOption &config) : ::aslp::SodWorker(WorldComm(), config) {} };
class EasgdWorker : public ::aslp::EasgdWorker { public: explicit EasgdWorker(float alpha) : ::aslp::EasgdWorker(ServedWorkerComm("easgd"), alpha) {} };
class AsgdWorker : public ::aslp::AsgdWorker { public: AsgdWorker() : ::aslp::AsgdWorker(ServedWorkerComm("asgd / masgd")) {} };
class EasgdServer : public ::aslp::EasgdServer { public: explicit EasgdServer(float alpha) : ::aslp::EasgdServer(WorldComm(), alpha) {} };
class AsgdServer : public ::aslp::AsgdServer { public: AsgdServer(float alpha, int sync_period) : ::aslp::AsgdServer(WorldComm(), alpha, sync_period) {} };
/* masgd-server.h: per-worker momentum buffers, no moving rate of its own */
class MasgdServer : public ::aslp::AsgdServer { public: MasgdServer(int sync_period, float momentum) : ::aslp::AsgdServer(WorldComm(), 1.0f, sync_period, true, momentum) {} };
/* nnet-mpi-sync.h: the two-rank synchroniser inside aslp-nnet-train-simple-mpi */
class NnetMpiSync : public ::aslp::PairSync { public: NnetMpiSync() : ::aslp::PairSync(WorldComm()) {} };

}  // namespace aslp_nnet
}  // namespace kaldi

#endif  /* ASLP_COMPAT_KALDI_PARALLEL_H_ */
