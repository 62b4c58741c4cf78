// capi.cpp -- C handle API (include/aslp_parallel.h) over Comm and the workers.
#include <cstring>
#include <memory>

#include "aslp_parallel.h"
#include "workers.h"

using namespace aslp;

struct aslp_comm_s {
  std::unique_ptr<Comm> comm;
};
struct aslp_worker_s {
  std::unique_ptr<IWorker> worker;
};

static thread_local std::string t_err;
#define API_BEGIN try {
#define API_END                      \
  }                                  \
  catch (const std::exception &e) {  \
    t_err = e.what();                \
    return 1;                        \
  }                                  \
  return 0;

static std::vector<std::pair<BaseFloat *, int>> Params(float *const *ptrs, const int *sizes, int n) {
  std::vector<std::pair<BaseFloat *, int>> p;
  for (int i = 0; i < n; i++) p.push_back(std::make_pair(ptrs[i], sizes[i]));
  return p;
}

extern "C" {

const char *aslp_parallel_last_error(void) { return t_err.c_str(); }

int aslp_comm_create_rccl(int rank, int num_nodes, const char *id_file, const char *token, int timeout_s, aslp_comm_t *out) {
  API_BEGIN
  RankFromEnvironment(&rank, &num_nodes);
  aslp_comm_s *h = new aslp_comm_s();
  try {
    h->comm.reset(NewRcclComm(rank, num_nodes, id_file ? id_file : "", timeout_s > 0 ? timeout_s : 900, token ? token : ""));
  } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int aslp_comm_create_shm(int rank, int num_nodes, const char *id_file, const char *token, int timeout_s, aslp_comm_t *out) {
  API_BEGIN
  RankFromEnvironment(&rank, &num_nodes);
  aslp_comm_s *h = new aslp_comm_s();
  try {
    h->comm.reset(NewShmComm(rank, num_nodes, id_file ? id_file : "", timeout_s > 0 ? timeout_s : 900, token ? token : ""));
  } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
void aslp_comm_free(aslp_comm_t c) { delete c; }
int aslp_comm_rank(aslp_comm_t c) { return c->comm->Rank(); }
int aslp_comm_num_nodes(aslp_comm_t c) { return c->comm->NumNodes(); }
int aslp_comm_barrier(aslp_comm_t c) { API_BEGIN c->comm->Barrier(); API_END }
int aslp_comm_allreduce_sum_f32(aslp_comm_t c, float *dev, size_t n) { API_BEGIN c->comm->AllReduceSum(dev, n); API_END }
int aslp_comm_allreduce_sum_f64(aslp_comm_t c, double *dev, size_t n) { API_BEGIN c->comm->AllReduceSum(dev, n); API_END }
int aslp_comm_allreduce_sum_host_i32(aslp_comm_t c, int32_t *host, size_t n) { API_BEGIN c->comm->AllReduceSumHost(host, n); API_END }
int aslp_comm_allreduce_sum_host_f64(aslp_comm_t c, double *host, size_t n) { API_BEGIN c->comm->AllReduceSumHost(host, n); API_END }
int aslp_comm_send_f32(aslp_comm_t c, int peer, float *dev, size_t n) {
  API_BEGIN c->comm->Send(peer, Comm::Buffers(1, std::make_pair(dev, (int)n))); API_END
}
int aslp_comm_recv_f32(aslp_comm_t c, int peer, float *dev, size_t n) {
  API_BEGIN c->comm->Recv(peer, Comm::Buffers(1, std::make_pair(dev, (int)n))); API_END
}
int aslp_comm_exchange_f32(aslp_comm_t c, int peer, float *send_dev, float *recv_dev, size_t n) {
  API_BEGIN
  c->comm->Exchange(peer, Comm::Buffers(1, std::make_pair(send_dev, (int)n)), Comm::Buffers(1, std::make_pair(recv_dev, (int)n)));
  API_END
}

int aslp_worker_create(aslp_comm_t c, const char *kind, float p0, float p1, aslp_worker_t *out) {
  API_BEGIN
  const std::string k = kind ? kind : "";
  aslp_worker_s *h = new aslp_worker_s();
  if (k == "bsp") h->worker.reset(new BspWorker(c->comm.get()));
  else if (k == "bmuf") h->worker.reset(new BmufWorker(c->comm.get(), p0, p1));
  else if (k == "easgd") h->worker.reset(new EasgdWorker(c->comm.get(), p0));
  else if (k == "asgd") h->worker.reset(new AsgdWorker(c->comm.get()));
  else { delete h; ASLP_ERR << "aslp_worker_create: unknown worker kind '" << k << "' (bsp | bmuf | easgd | asgd)"; }
  *out = h;
  API_END
}
void aslp_worker_free(aslp_worker_t w) { delete w; }
int aslp_worker_init_param(aslp_worker_t w, float *const *dev_ptrs, const int *sizes, int n) {
  API_BEGIN w->worker->InitParam(Params(dev_ptrs, sizes, n)); API_END
}
int aslp_comm_ranks_seen(aslp_comm_t c) { try { return c->comm->RanksSeen(); } catch (const std::exception &e) { t_err = e.what(); return -1; } }
const char *aslp_comm_transport(aslp_comm_t c) { return c->comm->Transport(); }
int aslp_worker_init_param_nnet(aslp_worker_t w, aslp_nnet_t net) {
  API_BEGIN
  const int n = aslp_nnet_get_gpu_params(net, nullptr, nullptr, 0);
  if (n < 0) ASLP_ERR << "aslp_worker_init_param_nnet: " << aslp_nnet_last_error();
  std::vector<float *> ptrs(n > 0 ? n : 1);
  std::vector<int> sizes(n > 0 ? n : 1);
  if (aslp_nnet_get_gpu_params(net, ptrs.data(), sizes.data(), n) != n) ASLP_ERR << "aslp_worker_init_param_nnet: " << aslp_nnet_last_error();
  w->worker->InitParam(Params(ptrs.data(), sizes.data(), n));
  if (aslp_nnet_param_writers_announce(net) != 0) ASLP_ERR << "aslp_worker_init_param_nnet: " << aslp_nnet_last_error();   // the workers call aslp_params_changed()
  API_END
}
int aslp_worker_synchronize(aslp_worker_t w, int num_worker_samples, int *more) {
  API_BEGIN
  const bool m = w->worker->Synchronize(num_worker_samples);
  if (more) *more = m ? 1 : 0;
  API_END
}
int aslp_worker_stop(aslp_worker_t w) { API_BEGIN w->worker->Stop(); API_END }

int aslp_server_run(aslp_comm_t c, const char *kind, float p0, float p1, int sync_period, float *const *dev_ptrs, const int *sizes, int n) {
  API_BEGIN
  const std::string k = kind ? kind : "";
  std::unique_ptr<IServer> s;
  if (k == "easgd") s.reset(new EasgdServer(c->comm.get(), p0));
  else if (k == "asgd") s.reset(new AsgdServer(c->comm.get(), p0, sync_period));
  else if (k == "masgd") s.reset(new AsgdServer(c->comm.get(), p0, sync_period, true, p1));
  else ASLP_ERR << "aslp_server_run: unknown server kind '" << k << "' (easgd | asgd | masgd)";
  s->InitParam(Params(dev_ptrs, sizes, n));
  s->Run();
  API_END
}

}  // extern "C"
