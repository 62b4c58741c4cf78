// aslp-parallel-selftest -- runs BspWorker / BmufWorker / SodWorker with N ranks as threads of this process (ThreadComm) on raw device
// buffers and prints every rank's parameters after each synchronisation, one line per (step, rank): the GPU test compares
// them with the closed forms of bsp-worker.cc:33-65 / bmuf-worker.cc:37-68 computed in numpy.
// (sod: of optimizer.h:40-171 applied to the summed deltas, sod-worker.cc:46-58).
// easgd | asgd | masgd: rank 0 is the parameter server, the other ranks are workers that take turns (worker 1, 2, ..., 1, 2, ...)
// so that the arrival order at the server -- which decides the result -- is the same in every run; [lr momentum] are then
// alpha (easgd, asgd) / the masgd momentum, and the asgd / masgd sync period.
// pair: PairSync with two ranks, one line per (exchange, rank).
// Usage: aslp-parallel-selftest <bsp|bmuf|sod:SOLVER|easgd|asgd|masgd|pair> <num-ranks> <dim> <steps> [bmuf-lr bmuf-momentum]   (sod runs the solver's defaults)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "common.h"
#include "workers.h"

using namespace aslp;

int main(int argc, char **argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s <bsp|bmuf> <num-ranks> <dim> <steps> [lr momentum]\n", argv[0]); return 1; }
  const std::string type = argv[1];
  const int N = atoi(argv[2]), dim = atoi(argv[3]), steps = atoi(argv[4]);
  const float lr = argc > 5 ? atof(argv[5]) : 1.0f, mom = argc > 6 ? atof(argv[6]) : 0.9f;
  auto group = NewThreadCommGroup(N);
  std::vector<std::vector<float>> out((size_t)N * (steps + 1));
  std::vector<std::string> errors(N);
  std::vector<std::thread> th;
  const bool served = type == "easgd" || type == "asgd" || type == "masgd";
  std::atomic<int> turn(0);  // served protocols: which (step, worker) may synchronise next
  for (int r = 0; r < N; r++) {
    th.emplace_back([&, r] {
      try {
        hipStream_t st;
        if (hipStreamCreate(&st) != hipSuccess) throw std::runtime_error("hipStreamCreate");
        set_cur_stream(st);
        std::unique_ptr<Comm> comm(NewThreadComm(group, r));
        if (type == "pair") {  // PairSync (aslp-nnet-train-simple-mpi): rank 1 runs out of data two exchanges before rank 0
          std::vector<float> h(dim);
          for (int i = 0; i < dim; i++) h[i] = 1 + 0.01f * i;
          float *d = nullptr;
          if (hipMalloc(&d, sizeof(float) * dim) != hipSuccess) throw std::runtime_error("hipMalloc");
          (void)hipMemcpyAsync(d, h.data(), sizeof(float) * dim, hipMemcpyHostToDevice, st);
          (void)hipStreamSynchronize(st);
          PairSync pair(comm.get());
          pair.Init({{d, dim / 2}, {d + dim / 2, dim - dim / 2}});
          const int mine = r == 0 ? steps : steps - 2;
          int k = 0;
          auto record = [&] {
            (void)hipMemcpyAsync(h.data(), d, sizeof(float) * dim, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            if (k <= steps) out[(size_t)k * N + r] = h;
            k++;
          };
          for (int s = 0; s < mine; s++) {
            (void)hipMemcpyAsync(h.data(), d, sizeof(float) * dim, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            for (float &x : h) x += 0.5f * (r + 1) + 0.25f * s;
            (void)hipMemcpyAsync(d, h.data(), sizeof(float) * dim, hipMemcpyHostToDevice, st);
            (void)hipStreamSynchronize(st);
            pair.Sync();
            record();
          }
          pair.SetSelfDone();
          do { pair.Sync(); record(); } while (!pair.AllDone());
          (void)hipFree(d);
          return;
        }
        if (served) {
          // every rank starts from the same model w[i] = 1 + 0.01 i; worker r adds 0.5 r + 0.25 s before its s-th exchange
          std::vector<float> h(dim);
          for (int i = 0; i < dim; i++) h[i] = 1 + 0.01f * i;
          float *d = nullptr;
          if (hipMalloc(&d, sizeof(float) * dim) != hipSuccess) throw std::runtime_error("hipMalloc");
          (void)hipMemcpyAsync(d, h.data(), sizeof(float) * dim, hipMemcpyHostToDevice, st);
          (void)hipStreamSynchronize(st);
          std::vector<std::pair<float *, int>> params = {{d, dim / 2}, {d + dim / 2, dim - dim / 2}};
          if (r == 0) {
            std::unique_ptr<IServer> srv;
            if (type == "easgd") srv.reset(new EasgdServer(comm.get(), lr));
            else if (type == "asgd") srv.reset(new AsgdServer(comm.get(), lr, (int)mom));
            else srv.reset(new AsgdServer(comm.get(), 1.0f, (int)mom, true, lr));
            srv->InitParam(params);
            srv->Run();
          } else {
            std::unique_ptr<IWorker> w;
            if (type == "easgd") w.reset(new EasgdWorker(comm.get(), lr));
            else w.reset(new AsgdWorker(comm.get()));
            w->InitParam(params);
            for (int s = 0; s < steps; s++) {
              (void)hipMemcpyAsync(h.data(), d, sizeof(float) * dim, hipMemcpyDeviceToHost, st);
              (void)hipStreamSynchronize(st);
              for (float &x : h) x += 0.5f * r + 0.25f * s;
              (void)hipMemcpyAsync(d, h.data(), sizeof(float) * dim, hipMemcpyHostToDevice, st);
              (void)hipStreamSynchronize(st);
              const int my_turn = s * (N - 1) + (r - 1);
              while (turn.load() != my_turn) std::this_thread::yield();
              w->Synchronize(100);
              (void)hipMemcpyAsync(h.data(), d, sizeof(float) * dim, hipMemcpyDeviceToHost, st);
              (void)hipStreamSynchronize(st);
              out[(size_t)s * N + r] = h;
              turn.fetch_add(1);
            }
            w->Stop();
          }
          (void)hipMemcpyAsync(h.data(), d, sizeof(float) * dim, hipMemcpyDeviceToHost, st);
          (void)hipStreamSynchronize(st);
          if (r == 0) out[(size_t)steps * N + r] = h;  // the server's final model
          (void)hipFree(d);
          return;
        }
        // two tensors (dim and dim / 2 + 1 floats): w[i] = rank + 1 + 0.01 i at the start
        const int n1 = dim, n2 = dim / 2 + 1;
        std::vector<float> h(n1 + n2);
        for (int i = 0; i < n1 + n2; i++) h[i] = r + 1 + 0.01f * i;
        float *d = nullptr;
        if (hipMalloc(&d, sizeof(float) * (n1 + n2)) != hipSuccess) throw std::runtime_error("hipMalloc");
        (void)hipMemcpyAsync(d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice, st);
        std::vector<std::pair<float *, int>> params = {{d, n1}, {d + n1, n2}};
        std::unique_ptr<IWorker> w;
        if (type == "bsp") w.reset(new BspWorker(comm.get()));
        else if (type.compare(0, 4, "sod:") == 0) { OptimizerOption o; o.solver = type.substr(4); w.reset(new SodWorker(comm.get(), o)); }
        else w.reset(new BmufWorker(comm.get(), lr, mom));
        w->InitParam(params);
        for (int s = 0; s <= steps; s++) {
          // "training": every rank moves its model by a rank- and step-dependent amount, then synchronises with a
          // rank-dependent sample count; rank r runs out of data after steps - r steps and calls Stop()
          const bool has_data = s < steps - r;
          if (has_data) {
            (void)hipMemcpyAsync(h.data(), d, sizeof(float) * h.size(), hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            for (size_t i = 0; i < h.size(); i++) h[i] += 0.5f * (r + 1) + 0.25f * s;
            (void)hipMemcpyAsync(d, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice, st);
            if (!w->Synchronize(100 * (r + 1) + s)) throw std::runtime_error("Synchronize returned false while data remains");
          } else {
            w->Stop();
            break;
          }
          (void)hipMemcpyAsync(h.data(), d, sizeof(float) * h.size(), hipMemcpyDeviceToHost, st);
          (void)hipStreamSynchronize(st);
          out[(size_t)s * N + r] = h;
        }
        (void)hipMemcpyAsync(h.data(), d, sizeof(float) * h.size(), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        out[(size_t)steps * N + r] = h;  // the final model of this rank
        w.reset();
        (void)hipFree(d);
      } catch (const std::exception &e) {
        errors[r] = e.what();
      }
    });
  }
  for (auto &t : th) t.join();
  for (int r = 0; r < N; r++)
    if (!errors[r].empty()) { std::fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str()); return 2; }
  for (int s = 0; s <= steps; s++)
    for (int r = 0; r < N; r++) {
      const auto &v = out[(size_t)s * N + r];
      if (v.empty()) continue;
      std::printf("%d %d", s, r);
      for (float x : v) std::printf(" %.9g", x);
      std::printf("\n");
    }
  return 0;
}
