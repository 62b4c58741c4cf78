// comm.cpp -- see comm.h.
#include "comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <ctime>
#include <mutex>
#include <cstdio>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <thread>

#include "aslp_kernels.h"
#include "common.h"

namespace aslp {

static void Hip(hipError_t e, const char *what) { if (e != hipSuccess) ASLP_ERR << what << ": " << hipGetErrorString(e); }
static void Nccl(ncclResult_t r, const char *what) { if (r != ncclSuccess) ASLP_ERR << what << ": " << ncclGetErrorString(r); }

void RankFromEnvironment(int *rank, int *num_nodes) {
  const char *rk[] = {"OMPI_COMM_WORLD_RANK", "PMI_RANK", "RANK"}, *sz[] = {"OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "WORLD_SIZE"};
  if (*rank < 0)
    for (const char *k : rk) if (getenv(k)) { *rank = atoi(getenv(k)); break; }
  if (*num_nodes <= 0)
    for (const char *k : sz) if (getenv(k)) { *num_nodes = atoi(getenv(k)); break; }
  if (*rank < 0) *rank = 0;
  if (*num_nodes <= 0) *num_nodes = 1;
  if (*rank >= *num_nodes) ASLP_ERR << "rank " << *rank << " outside a group of " << *num_nodes;
}

// ---- one process per rank: rendezvous through a file, arrival order through a named pipe --------------------------
namespace {
// The rendezvous file is bound to ONE launch: {magic, world size, token, 128 bytes of payload}.  The payload is the transport's
// (RcclComm: the ncclUniqueId; ShmComm: the name of the shared-memory segment), the magic names the transport, so ranks started
// with different transports never join each other.  The token comes from the caller,
// else ASLP_COMM_TOKEN, else the launcher's job id (torchrun, Slurm, PMI, and the PMIx / Open MPI ids mpirun exports -- the
// reference's recipes start the workers with mpirun): every rank of a launch sees the same value, a file left behind by a run
// that died between writing and removing it carries another one and is ignored.  With no token at all a leftover is told
// apart by its age (ASLP_COMM_MAX_AGE_S, default 600 s) and by a second look: a record is only taken once it has stayed
// unchanged for a grace period, within which a rank 0 that is starting up has replaced it.  Rank 0 removes whatever is there
// FIRST THING (before it forms the payload: ncclGetUniqueId can take minutes on a cold box), so the window in which the others
// can see a leftover closes as soon as rank 0 runs.
class FileRendezvousComm : public Comm {
 public:
  struct IdRecord {
    char magic[8];
    int32 world, reserved;
    char token[64];
    unsigned char payload[128];
  };
  static std::string LaunchToken(const std::string &given) {
    if (!given.empty()) return given;
    for (const char *k : {"ASLP_COMM_TOKEN", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "PMI_ID", "PMIX_NAMESPACE", "OMPI_MCA_ess_base_jobid",
                          "OMPI_MCA_orte_ess_jobid", "PMI_JOBID"})
      if (getenv(k) && getenv(k)[0]) return getenv(k);
    return std::string();
  }
  int Rank() const { return rank_; }
  int NumNodes() const { return n_; }
  // the MPI_Recv(MPI_ANY_SOURCE) of easgd-server.cc:44 as a named pipe next to the rendezvous file
  void PostToServer(int32 msg) {
    const int32 rec[2] = {rank_, msg};  // 8 bytes: one atomic write (< PIPE_BUF), so messages of different workers never mix
    if (ctl_fd_ < 0) ASLP_ERR << "this group has no control pipe (" << ctl_path_ << "): the served protocols need one";
    if (write(ctl_fd_, rec, sizeof(rec)) != (ssize_t)sizeof(rec)) ASLP_ERR << "control pipe write failed: " << strerror(errno);
  }
  void WaitFromWorker(int *src, int32 *msg) {
    if (ctl_fd_ < 0) ASLP_ERR << "this group has no control pipe (" << ctl_path_ << "): the served protocols need one";
    int32 rec[2];
    size_t got = 0;
    while (got < sizeof(rec)) {
      const ssize_t k = read(ctl_fd_, reinterpret_cast<char *>(rec) + got, sizeof(rec) - got);
      if (k < 0 && errno == EINTR) continue;
      if (k <= 0) ASLP_ERR << "control pipe read failed: " << strerror(errno);
      got += (size_t)k;
    }
    *src = rec[0];
    *msg = rec[1];
  }

 protected:
  FileRendezvousComm(int rank, int n) : rank_(rank), n_(n), ctl_fd_(-1) {}
  ~FileRendezvousComm() {
    if (ctl_fd_ >= 0) (void)close(ctl_fd_);
    if (rank_ == 0 && n_ > 1 && !ctl_path_.empty()) (void)unlink(ctl_path_.c_str());
  }
  // rank 0: remove leftovers, form the payload (make_payload), create the control pipe, publish the record atomically;
  // the others: wait (up to timeout_s) for this launch's record and open the pipe.  payload: 128 bytes, out on every rank.
  template <class MakePayload>
  void Join(const char *magic8, const std::string &id_file, int timeout_s, const std::string &token_arg, const char *who, MakePayload make_payload,
            unsigned char *payload) {
    IdRecord rec;
    std::memset(&rec, 0, sizeof(rec));
    const std::string token = LaunchToken(token_arg);
    if (token.size() >= sizeof(rec.token)) ASLP_ERR << who << ": launch token longer than " << sizeof(rec.token) - 1 << " characters";
    if (n_ > 1 && id_file.empty()) ASLP_ERR << who << ": more than one rank needs a rendezvous file (--comm-file)";
    id_file_ = id_file;
    ctl_path_ = id_file.empty() ? std::string() : id_file + ".ctl";
    const double max_age = getenv("ASLP_COMM_MAX_AGE_S") ? atof(getenv("ASLP_COMM_MAX_AGE_S")) : 600.0;
    const time_t started = time(nullptr);
    if (rank_ == 0) {
      if (n_ > 1) {   // leftovers of a run that died go before anything slow happens: nobody may join them
        (void)unlink(id_file.c_str());
        (void)unlink(ctl_path_.c_str());
      }
      make_payload(rec.payload);
      if (n_ > 1) {
        // the control pipe exists before the record does, so whoever found the record can open it; O_RDWR: never sees end-of-file
        // (only the served protocols need it: on a filesystem without FIFOs the collective workers still run)
        if (mkfifo(ctl_path_.c_str(), 0600) == 0) ctl_fd_ = open(ctl_path_.c_str(), O_RDWR);
        if (ctl_fd_ < 0) ASLP_WARN << "no control pipe at " << ctl_path_ << " (" << strerror(errno) << "): easgd / asgd / masgd are unavailable in this group";
        std::memcpy(rec.magic, magic8, 8);
        rec.world = n_;
        std::strncpy(rec.token, token.c_str(), sizeof(rec.token) - 1);
        const std::string tmp = id_file + ".tmp." + std::to_string((long)getpid());
        { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char *>(&rec), sizeof(rec)); if (!f.good()) ASLP_ERR << "cannot write " << tmp; }
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) ASLP_ERR << "cannot create " << id_file;
      }
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      std::string why = "no file yet";
      while (true) {
        struct stat st;
        std::ifstream f(id_file, std::ios::binary);
        if (f.good() && stat(id_file.c_str(), &st) == 0) {
          f.read(reinterpret_cast<char *>(&rec), sizeof(rec));
          rec.token[sizeof(rec.token) - 1] = 0;
          if (f.gcount() != (std::streamsize)sizeof(rec) || std::memcmp(rec.magic, magic8, 8) != 0) why = "not a rendezvous record of this transport";
          else if (rec.world != n_) why = "written for a group of " + std::to_string(rec.world);
          else if (token != rec.token) why = "belongs to another launch (token '" + std::string(rec.token) + "')";
          else if (difftime(started, st.st_mtime) > max_age) why = "older than " + std::to_string((int)max_age) + " s (a leftover; see ASLP_COMM_MAX_AGE_S)";
          else if (!token.empty()) break;   // this launch's own record
          else {
            // no token: take the record only if it is still the same after a grace period (a rank 0 that is just starting has
            // removed or replaced a leftover by then)
            static const double grace = getenv("ASLP_COMM_GRACE_S") ? atof(getenv("ASLP_COMM_GRACE_S")) : 2.0;
            std::this_thread::sleep_for(std::chrono::milliseconds((long)(grace * 1000)));
            IdRecord again;
            std::memset(&again, 0, sizeof(again));
            std::ifstream f2(id_file, std::ios::binary);
            f2.read(reinterpret_cast<char *>(&again), sizeof(again));
            again.token[sizeof(again.token) - 1] = 0;
            if (f2.gcount() == (std::streamsize)sizeof(again) && std::memcmp(&again, &rec, sizeof(rec)) == 0) break;
            why = "changed while it was being read (a leftover replaced by this launch's rank 0)";
            continue;   // look again at once: the fresh record may already be there
          }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
          ASLP_ERR << who << ": rank " << rank_ << " timed out waiting for " << id_file << " (" << why << ")";
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
    }
    if (rank_ != 0 && n_ > 1) ctl_fd_ = open(ctl_path_.c_str(), O_WRONLY | O_NONBLOCK);  // rank 0 holds the read end already; -1: see PostToServer
    std::memcpy(payload, rec.payload, sizeof(rec.payload));
  }
  // everybody has joined: the next run writes a fresh record
  void Joined() { if (rank_ == 0 && n_ > 1) std::remove(id_file_.c_str()); }
  int rank_, n_;
  std::string id_file_, ctl_path_;
  int ctl_fd_;
};

// ---- RCCL ---------------------------------------------------------------------------------------------------
class RcclComm : public FileRendezvousComm {
 public:
  RcclComm(int rank, int n, const std::string &id_file, int timeout_s, const std::string &token_arg)
      : FileRendezvousComm(rank, n), comm_(nullptr), scratch_(nullptr), scratch_bytes_(0) {
    static_assert(sizeof(ncclUniqueId) <= sizeof(IdRecord().payload), "ncclUniqueId larger than the rendezvous payload");
    ncclUniqueId id;
    unsigned char payload[128];
    Join("ASLPRCCL", id_file, timeout_s, token_arg, "RcclComm", [](unsigned char *pl) {
      ncclUniqueId mine;
      Nccl(ncclGetUniqueId(&mine), "ncclGetUniqueId");
      std::memcpy(pl, &mine, sizeof(mine));
    }, payload);
    std::memcpy(&id, payload, sizeof(id));
    InitWithTimeout(id, timeout_s);
    Barrier();
    Joined();
  }
  void InitWithTimeout(const ncclUniqueId &id, int timeout_s) {
    int device = 0;
    Hip(hipGetDevice(&device), "hipGetDevice");
    struct Shared { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
    auto sh = std::make_shared<Shared>();
    const int rank = rank_, n = n_;
    std::thread([sh, id, device, rank, n] {  // detached: if it never returns, the caller has long thrown
      (void)hipSetDevice(device);
      ncclComm_t c = nullptr;
      const ncclResult_t r = ncclCommInitRank(&c, n, id, rank);
      std::lock_guard<std::mutex> lk(sh->mu);
      sh->res = r; sh->comm = c; sh->done = true;
      sh->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(sh->mu);
    if (!sh->cv.wait_for(lk, std::chrono::seconds(timeout_s > 0 ? timeout_s : 900), [&] { return sh->done; }))
      ASLP_ERR << "RcclComm: rank " << rank_ << " of " << n_ << ": ncclCommInitRank did not return within " << timeout_s
               << " s (another rank missing, or a rendezvous id of another run)";
    Nccl(sh->res, "ncclCommInitRank");
    comm_ = sh->comm;
  }
  ~RcclComm() {
    if (scratch_) (void)hipFree(scratch_);
    if (comm_) (void)ncclCommDestroy(comm_);
  }
  int RanksSeen() const {
    int n = 0;
    Nccl(ncclCommCount(comm_, &n), "ncclCommCount");
    return n;
  }
  const char *Transport() const { return "rccl"; }
  void Send(int peer, const Buffers &bufs) { P2P(peer, &bufs, nullptr); }
  void Recv(int peer, const Buffers &bufs) { P2P(peer, nullptr, &bufs); }
  void Exchange(int peer, const Buffers &send, const Buffers &recv) { P2P(peer, &send, &recv); }
  void Barrier() {
    int32 one = 1;
    AllReduceSumHost(&one, 1);
  }
  void AllReduceSum(float *dev, size_t n) {
    if (n) Nccl(ncclAllReduce(dev, dev, n, ncclFloat, ncclSum, comm_, cur_stream()), "ncclAllReduce(float)");
  }
  void AllReduceSum(double *dev, size_t n) {
    if (n) Nccl(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm_, cur_stream()), "ncclAllReduce(double)");
  }
  void AllReduceSumMany(const std::vector<std::pair<float *, int>> &bufs) {
    // tensors that lie back to back in memory (a parameter arena, the pitched rows of one matrix followed by its bias) travel as ONE
    // collective; whatever is left goes into one group
    std::vector<std::pair<float *, size_t>> spans;
    for (auto &b : bufs) {
      if (b.second <= 0) continue;
      if (!spans.empty() && spans.back().first + spans.back().second == b.first) spans.back().second += (size_t)b.second;
      else spans.emplace_back(b.first, (size_t)b.second);
    }
    if (spans.size() > 1) Nccl(ncclGroupStart(), "ncclGroupStart");
    for (auto &sp : spans) Nccl(ncclAllReduce(sp.first, sp.first, sp.second, ncclFloat, ncclSum, comm_, cur_stream()), "ncclAllReduce(float)");
    if (spans.size() > 1) Nccl(ncclGroupEnd(), "ncclGroupEnd");
  }
  void AllReduceSumHost(int32 *host, size_t n) { HostReduce(host, n, ncclInt32); }
  void AllReduceSumHost(double *host, size_t n) { HostReduce(host, n, ncclDouble); }

 private:
  void P2P(int peer, const Buffers *send, const Buffers *recv) {
    Nccl(ncclGroupStart(), "ncclGroupStart");
    if (send)
      for (auto &b : *send)
        if (b.second > 0) Nccl(ncclSend(b.first, (size_t)b.second, ncclFloat, peer, comm_, cur_stream()), "ncclSend");
    if (recv)
      for (auto &b : *recv)
        if (b.second > 0) Nccl(ncclRecv(b.first, (size_t)b.second, ncclFloat, peer, comm_, cur_stream()), "ncclRecv");
    Nccl(ncclGroupEnd(), "ncclGroupEnd");
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
  }
  template <class T>
  void HostReduce(T *host, size_t n, ncclDataType_t type) {
    if (!n) return;
    const size_t bytes = sizeof(T) * n;
    if (bytes > scratch_bytes_) {
      if (scratch_) Hip(hipFree(scratch_), "hipFree");
      Hip(hipMalloc(&scratch_, bytes < 256 ? 256 : bytes), "hipMalloc");
      scratch_bytes_ = bytes < 256 ? 256 : bytes;
    }
    Hip(hipMemcpyAsync(scratch_, host, bytes, hipMemcpyHostToDevice, cur_stream()), "hipMemcpy H2D");
    Nccl(ncclAllReduce(scratch_, scratch_, n, type, ncclSum, comm_, cur_stream()), "ncclAllReduce(host value)");
    Hip(hipMemcpyAsync(host, scratch_, bytes, hipMemcpyDeviceToHost, cur_stream()), "hipMemcpy D2H");
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
  }
  ncclComm_t comm_;
  void *scratch_;
  size_t scratch_bytes_;
};

// ---- processes that share GPUs: tensors staged through one shared-memory segment ----------------------------------------
// RCCL refuses two ranks on one device, so a one-GPU box cannot run the product's process-per-rank machinery -- rendezvous record,
// control pipe, arrival order at the server, the workers' collectives from separate address spaces -- on RcclComm.  ShmComm is the
// same Comm for ranks that are separate OS processes on ANY devices (all on one GPU included): behind the same rendezvous file and
// the same control pipe, with the tensors moved device -> host slot -> device through a POSIX shared-memory segment that rank 0
// creates and names in the rendezvous payload.  One slot per rank:
//   all-reduce   every rank copies a chunk of its buffer into its slot, barrier, every rank adds the N slots in RANK ORDER (so all
//                ranks, and every run, form the same bits), barrier, copies the sum back
//   send / recv  the sender's slot is a mailbox addressed to one receiver (flag = receiver + 1), chunk by chunk
// Host-staged, so it is a functional transport (tests, single-GPU development), not the one a multi-GPU node should train on.
class ShmComm : public FileRendezvousComm {
 public:
  static constexpr int kMaxRanks = 64;
  struct Header {
    std::atomic<int> arrived, generation;
    std::atomic<int> flag[kMaxRanks];        // mailbox state of rank r's slot: 0 free, d + 1 = holds a chunk for rank d
    std::atomic<long> aborted;               // a rank that gave up (timeout) sets it: the others stop waiting too
  };
  ShmComm(int rank, int n, const std::string &id_file, int timeout_s, const std::string &token_arg)
      : FileRendezvousComm(rank, n), base_(nullptr), bytes_(0), hdr_(nullptr), timeout_s_(timeout_s > 0 ? timeout_s : 900) {
    if (n > kMaxRanks) ASLP_ERR << "ShmComm: at most " << kMaxRanks << " ranks";
    // ranks of this transport may share a GPU: the persistent LSTM / GRU launches of libaslp_hip then take a cross-process device lock
    // (include/aslp_kernels.h aslp_device_shared; read from the environment at the first such launch).  An explicit setting wins.
    // The switch is process-wide and stays on after this communicator is gone: another one may live in the process, and a lock around a launch
    // that no longer needs it costs microseconds.
    if (n > 1) {
      const char *given = getenv("ASLP_DEVICE_SHARED");
      if (given == nullptr) (void)setenv("ASLP_DEVICE_SHARED", "1", 0);   // (for child processes; a failed setenv only loses that)
      if (given == nullptr || given[0] == '1') aslp_device_shared(1);       // also when libaslp_hip has already read the environment
    }
    static_assert(std::atomic<int>::is_always_lock_free && std::atomic<long>::is_always_lock_free, "address-free atomics needed in shared memory");
    const size_t slot_mb = getenv("ASLP_SHM_SLOT_MB") ? (size_t)atol(getenv("ASLP_SHM_SLOT_MB")) : 16;
    slot_bytes_ = (slot_mb < 1 ? 1 : slot_mb) << 20;
    bytes_ = kHeaderBytes + (size_t)n * slot_bytes_;
    unsigned char payload[128];
    Join("ASLPSHM1", id_file, timeout_s, token_arg, "ShmComm", [&](unsigned char *pl) {
      // rank 0 creates the segment (a fresh name per launch) and zeroes its header before anybody can learn the name
      char name[64];
      std::snprintf(name, sizeof(name), "/aslp_shm_%ld_%lx", (long)getpid(), (unsigned long)std::chrono::steady_clock::now().time_since_epoch().count());
      const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) ASLP_ERR << "ShmComm: shm_open(" << name << "): " << strerror(errno);
      if (ftruncate(fd, (off_t)bytes_) != 0) { (void)close(fd); (void)shm_unlink(name); ASLP_ERR << "ShmComm: cannot size the segment to " << bytes_ << " bytes: " << strerror(errno); }
      Map(fd);
      (void)close(fd);
      std::memset(base_, 0, kHeaderBytes);
      name_ = name;
      std::memset(pl, 0, 128);
      std::memcpy(pl, name, std::strlen(name));
      std::memcpy(pl + 64, &slot_bytes_, sizeof(slot_bytes_));
    }, payload);
    if (rank != 0) {
      payload[63] = 0;
      name_ = reinterpret_cast<const char *>(payload);
      size_t their_slot = 0;
      std::memcpy(&their_slot, payload + 64, sizeof(their_slot));
      if (their_slot != slot_bytes_) ASLP_ERR << "ShmComm: rank 0 uses slots of " << their_slot << " bytes, this rank " << slot_bytes_ << " (ASLP_SHM_SLOT_MB differs)";
      const int fd = shm_open(name_.c_str(), O_RDWR, 0600);
      if (fd < 0) ASLP_ERR << "ShmComm: rank " << rank << " cannot open segment " << name_ << ": " << strerror(errno);
      Map(fd);
      (void)close(fd);
    }
    try {
      Barrier();
    } catch (...) {   // a peer never came: no destructor will run for this object, so name and mapping go here
      if (rank == 0) (void)shm_unlink(name_.c_str());
      if (base_) (void)munmap(base_, bytes_);
      base_ = nullptr;
      throw;
    }
    if (rank == 0) (void)shm_unlink(name_.c_str());   // everybody has it mapped: the name can go, the memory lives until the last unmap
    Joined();
  }
  const char *Transport() const { return "shm"; }
  ~ShmComm() {
    if (base_) (void)munmap(base_, bytes_);
  }
  void Barrier() {
    if (n_ == 1) return;
    const int gen = hdr_->generation.load(std::memory_order_acquire);
    if (hdr_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
      hdr_->arrived.store(0, std::memory_order_relaxed);
      hdr_->generation.fetch_add(1, std::memory_order_release);
    } else {
      WaitUntil([&] { return hdr_->generation.load(std::memory_order_acquire) != gen; }, "barrier");
    }
  }
  void AllReduceSum(float *dev, size_t n) { Reduce(dev, n, true); }
  void AllReduceSum(double *dev, size_t n) { Reduce(dev, n, true); }
  void AllReduceSumHost(int32 *host, size_t n) { Reduce(host, n, false); }
  void AllReduceSumHost(double *host, size_t n) { Reduce(host, n, false); }
  void Send(int peer, const Buffers &bufs) { Transfer(peer, &bufs, nullptr); }
  void Recv(int peer, const Buffers &bufs) { Transfer(peer, nullptr, &bufs); }
  void Exchange(int peer, const Buffers &send, const Buffers &recv) { Transfer(peer, &send, &recv); }

 private:
  static constexpr size_t kHeaderBytes = 4096;
  void Map(int fd) {
    base_ = static_cast<unsigned char *>(mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    if (base_ == MAP_FAILED) { base_ = nullptr; ASLP_ERR << "ShmComm: mmap of " << bytes_ << " bytes: " << strerror(errno); }
    static_assert(sizeof(Header) <= kHeaderBytes, "header does not fit");
    hdr_ = reinterpret_cast<Header *>(base_);
  }
  unsigned char *Slot(int r) const { return base_ + kHeaderBytes + (size_t)r * slot_bytes_; }
  template <class Pred>
  void WaitUntil(Pred done, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; !done(); spins++) {
      if (spins < 2000) { sched_yield(); continue; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spins & 1023u) == 0u) {
        if (hdr_->aborted.load(std::memory_order_relaxed) != 0) ASLP_ERR << "ShmComm: rank " << rank_ << ": another rank gave up (" << what << ")";
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) {
          hdr_->aborted.store(1, std::memory_order_relaxed);
          ASLP_ERR << "ShmComm: rank " << rank_ << " of " << n_ << " waited " << timeout_s_ << " s in " << what << " (a rank died?)";
        }
      }
    }
  }
  // in-place sum over the ranks of n elements at p (device memory if on_device, else host)
  template <class T>
  void Reduce(T *p, size_t n, bool on_device) {
    if (!n || n_ == 1) return;
    const size_t per = slot_bytes_ / sizeof(T);
    std::vector<T> acc;
    for (size_t o = 0; o < n; o += per) {
      const size_t m = n - o < per ? n - o : per;
      T *mine = reinterpret_cast<T *>(Slot(rank_));
      if (on_device) {
        Hip(hipMemcpyAsync(mine, p + o, sizeof(T) * m, hipMemcpyDeviceToHost, cur_stream()), "hipMemcpy D2H");
        Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
      } else {
        std::memcpy(mine, p + o, sizeof(T) * m);
      }
      Barrier();   // every slot holds its rank's chunk
      acc.assign(reinterpret_cast<const T *>(Slot(0)), reinterpret_cast<const T *>(Slot(0)) + m);
      for (int r = 1; r < n_; r++) {
        const T *q = reinterpret_cast<const T *>(Slot(r));
        for (size_t i = 0; i < m; i++) acc[i] += q[i];
      }
      Barrier();   // everybody has read every slot: they may be overwritten
      if (on_device) {
        Hip(hipMemcpyAsync(p + o, acc.data(), sizeof(T) * m, hipMemcpyHostToDevice, cur_stream()), "hipMemcpy H2D");
        Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
      } else {
        std::memcpy(p + o, acc.data(), sizeof(T) * m);
      }
    }
  }
  // a cursor over the chunks of a buffer list
  struct Cursor {
    const Buffers *b; size_t i, off;
    bool Done() const { return !b || i >= b->size(); }
    void Skip() { while (b && i < b->size() && (*b)[i].second <= 0) i++; }
  };
  // send and / or receive whole buffer lists to / from one peer, chunk by chunk; my slot carries what I send, the peer's what I receive
  void Transfer(int peer, const Buffers *send, const Buffers *recv) {
    if (peer < 0 || peer >= n_ || peer == rank_) ASLP_ERR << "ShmComm: bad peer " << peer;
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");   // what is sent is final
    Cursor sc{send, 0, 0}, rc{recv, 0, 0};
    sc.Skip(); rc.Skip();
    const size_t per = slot_bytes_ / sizeof(float);
    while (!sc.Done() || !rc.Done()) {
      bool progressed = false;
      if (!sc.Done() && hdr_->flag[rank_].load(std::memory_order_acquire) == 0) {
        const auto &buf = (*sc.b)[sc.i];
        const size_t m = (size_t)buf.second - sc.off < per ? (size_t)buf.second - sc.off : per;
        Hip(hipMemcpy(Slot(rank_), buf.first + sc.off, sizeof(float) * m, hipMemcpyDeviceToHost), "hipMemcpy D2H");
        hdr_->flag[rank_].store(peer + 1, std::memory_order_release);
        sc.off += m;
        if (sc.off >= (size_t)buf.second) { sc.i++; sc.off = 0; sc.Skip(); }
        progressed = true;
      }
      if (!rc.Done() && hdr_->flag[peer].load(std::memory_order_acquire) == rank_ + 1) {
        const auto &buf = (*rc.b)[rc.i];
        const size_t m = (size_t)buf.second - rc.off < per ? (size_t)buf.second - rc.off : per;
        Hip(hipMemcpy(buf.first + rc.off, Slot(peer), sizeof(float) * m, hipMemcpyHostToDevice), "hipMemcpy H2D");
        hdr_->flag[peer].store(0, std::memory_order_release);
        rc.off += m;
        if (rc.off >= (size_t)buf.second) { rc.i++; rc.off = 0; rc.Skip(); }
        progressed = true;
      }
      if (!progressed)
        WaitUntil([&] {
          return (!sc.Done() && hdr_->flag[rank_].load(std::memory_order_acquire) == 0) ||
                 (!rc.Done() && hdr_->flag[peer].load(std::memory_order_acquire) == rank_ + 1);
        }, "send / recv");
    }
    // synchronous send: what I sent has been taken before I return, so the slot is free for whatever uses it next (a collective writes
    // into it without looking at the mailbox flag)
    if (send) WaitUntil([&] { return hdr_->flag[rank_].load(std::memory_order_acquire) == 0; }, "send (waiting for the receiver)");
  }
  unsigned char *base_;
  size_t bytes_, slot_bytes_;
  Header *hdr_;
  std::string name_;
  int timeout_s_;
};
}  // namespace

Comm *NewShmComm(int rank, int num_nodes, const std::string &id_file, int timeout_s, const std::string &token) {
  return new ShmComm(rank, num_nodes, id_file, timeout_s, token);
}
// RcclComm unless ASLP_COMM_TRANSPORT=shm (or transport == "shm")
Comm *NewProcessComm(const std::string &transport, int rank, int num_nodes, const std::string &id_file, int timeout_s, const std::string &token) {
  std::string t = transport;
  if (t.empty() && getenv("ASLP_COMM_TRANSPORT")) t = getenv("ASLP_COMM_TRANSPORT");
  if (t.empty() || t == "rccl") return NewRcclComm(rank, num_nodes, id_file, timeout_s, token);
  if (t == "shm") return NewShmComm(rank, num_nodes, id_file, timeout_s, token);
  ASLP_ERR << "unknown communicator transport '" << t << "' (rccl | shm)";
  return nullptr;
}

Comm *NewRcclComm(int rank, int num_nodes, const std::string &id_file, int timeout_s, const std::string &token) {
  return new RcclComm(rank, num_nodes, id_file, timeout_s, token);
}

// ---- threads of one process ---------------------------------------------------------------------------------------
class ThreadCommGroup {
 public:
  explicit ThreadCommGroup(int n) : n(n), arrived(0), generation(0), ptrs(n, nullptr), hostsum_i(0), hostsum_d(0.0), tmp(nullptr), tmp_bytes(0) {}
  ~ThreadCommGroup() { if (tmp) (void)hipFree(tmp); }
  void Barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != gen; });
  }
  int n, arrived;
  long generation;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<void *> ptrs;
  std::vector<int32> host_i;
  std::vector<double> host_d;
  long long hostsum_i;
  double hostsum_d;
  void *tmp;
  size_t tmp_bytes;
  // server-based protocols: control queue (arrival order) and one rendezvous slot per ordered pair of ranks
  std::deque<std::pair<int, int32>> ctl;
  struct Slot { const Comm::Buffers *send = nullptr; bool taken = false; };
  std::map<std::pair<int, int>, Slot> slots;
};
std::shared_ptr<ThreadCommGroup> NewThreadCommGroup(int n) { return std::make_shared<ThreadCommGroup>(n); }

namespace {
__global__ void add_kernel_f(float *dst, const float *src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
__global__ void add_kernel_d(double *dst, const double *src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
class ThreadComm : public Comm {
 public:
  ThreadComm(std::shared_ptr<ThreadCommGroup> g, int rank) : g_(g), rank_(rank) {}
  int Rank() const { return rank_; }
  int NumNodes() const { return g_->n; }
  const char *Transport() const { return "threads"; }
  void Barrier() { g_->Barrier(); }
  void AllReduceSum(float *dev, size_t n) { Reduce(dev, n); }
  void AllReduceSum(double *dev, size_t n) { Reduce(dev, n); }
  void AllReduceSumHost(int32 *host, size_t n) { HostReduce(host, n, &g_->host_i); }
  void AllReduceSumHost(double *host, size_t n) { HostReduce(host, n, &g_->host_d); }
  void PostToServer(int32 msg) {
    std::lock_guard<std::mutex> lk(g_->mu);
    g_->ctl.push_back(std::make_pair(rank_, msg));
    g_->cv.notify_all();
  }
  void WaitFromWorker(int *src, int32 *msg) {
    std::unique_lock<std::mutex> lk(g_->mu);
    g_->cv.wait(lk, [&] { return !g_->ctl.empty(); });
    *src = g_->ctl.front().first;
    *msg = g_->ctl.front().second;
    g_->ctl.pop_front();
  }
  // rendezvous: the sender offers its buffers and waits until the receiver has copied them (device to device)
  void Send(int peer, const Buffers &bufs) {
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");  // what is sent is final
    std::unique_lock<std::mutex> lk(g_->mu);
    ThreadCommGroup::Slot &slot = g_->slots[std::make_pair(rank_, peer)];
    slot.send = &bufs;
    slot.taken = false;
    g_->cv.notify_all();
    g_->cv.wait(lk, [&] { return slot.taken; });
    slot.send = nullptr;
    slot.taken = false;
  }
  void Recv(int peer, const Buffers &bufs) {
    const Buffers *src = nullptr;
    {
      std::unique_lock<std::mutex> lk(g_->mu);
      ThreadCommGroup::Slot &slot = g_->slots[std::make_pair(peer, rank_)];
      g_->cv.wait(lk, [&] { return slot.send != nullptr && !slot.taken; });
      src = slot.send;
    }
    ASLP_ASSERT(src->size() == bufs.size());
    for (size_t i = 0; i < bufs.size(); i++) {
      ASLP_ASSERT((*src)[i].second == bufs[i].second);
      if (bufs[i].second > 0)
        Hip(hipMemcpyAsync(bufs[i].first, (*src)[i].first, sizeof(float) * (size_t)bufs[i].second, hipMemcpyDeviceToDevice, cur_stream()), "hipMemcpy D2D");
    }
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
    std::lock_guard<std::mutex> lk(g_->mu);
    g_->slots[std::make_pair(peer, rank_)].taken = true;
    g_->cv.notify_all();
  }
  void Exchange(int peer, const Buffers &send, const Buffers &recv) {
    if (rank_ < peer) { Send(peer, send); Recv(peer, recv); }
    else { Recv(peer, recv); Send(peer, send); }
  }

 private:
  template <class T>
  void HostReduce(T *host, size_t n, std::vector<T> *acc) {
    {
      std::lock_guard<std::mutex> lk(g_->mu);
      if (acc->size() != n) acc->assign(n, T(0));
    }
    g_->Barrier();
    {
      std::lock_guard<std::mutex> lk(g_->mu);
      for (size_t i = 0; i < n; i++) (*acc)[i] += host[i];
    }
    g_->Barrier();
    for (size_t i = 0; i < n; i++) host[i] = (*acc)[i];
    g_->Barrier();
    if (rank_ == 0) { std::lock_guard<std::mutex> lk(g_->mu); acc->clear(); }
    g_->Barrier();
  }
  static void Add(float *d, const float *s, size_t n) { hipLaunchKernelGGL(add_kernel_f, dim3(256), dim3(256), 0, cur_stream(), d, s, n); }
  static void Add(double *d, const double *s, size_t n) { hipLaunchKernelGGL(add_kernel_d, dim3(256), dim3(256), 0, cur_stream(), d, s, n); }
  template <class T>
  void Reduce(T *dev, size_t n) {
    if (!n) return;
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");  // this rank's buffer is final
    g_->ptrs[rank_] = dev;
    g_->Barrier();
    if (rank_ == 0) {  // sum in rank order into a temporary
      const size_t bytes = sizeof(T) * n;
      if (bytes > g_->tmp_bytes) { if (g_->tmp) Hip(hipFree(g_->tmp), "hipFree"); Hip(hipMalloc(&g_->tmp, bytes), "hipMalloc"); g_->tmp_bytes = bytes; }
      Hip(hipMemcpyAsync(g_->tmp, g_->ptrs[0], bytes, hipMemcpyDeviceToDevice, cur_stream()), "hipMemcpy D2D");
      for (int r = 1; r < g_->n; r++) Add(static_cast<T *>(g_->tmp), static_cast<const T *>(g_->ptrs[r]), n);
      Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
    }
    g_->Barrier();
    Hip(hipMemcpyAsync(dev, g_->tmp, sizeof(T) * n, hipMemcpyDeviceToDevice, cur_stream()), "hipMemcpy D2D");
    Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
    g_->Barrier();
  }
  std::shared_ptr<ThreadCommGroup> g_;
  int rank_;
};
}  // namespace
Comm *NewThreadComm(std::shared_ptr<ThreadCommGroup> group, int rank) { return new ThreadComm(group, rank); }

}  // namespace aslp
