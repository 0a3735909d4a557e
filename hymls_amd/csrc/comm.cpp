// comm.cpp -- see comm.hpp
#include "comm.hpp"
#include "device.hpp"

namespace hymls {

void Comm::a2a_host(const void* send, const std::vector<int64_t>& scnt, void* recv, const std::vector<int64_t>& rcnt,
                    int elem_bytes) const {
  HYMLS_CHECK(alltoallv != nullptr, -2, "sharded run without a transport (hymls_mi_set_comm)");
  const int ierr = alltoallv(ctx, send, scnt.data(), recv, rcnt.data(), elem_bytes, 0);
  HYMLS_CHECK(ierr == 0, -3, std::string("host all-to-all failed in the transport ") + rccl_last_error(*this));
}

std::vector<int64_t> Comm::exchange_counts(const std::vector<int64_t>& scnt) const {
  std::vector<int64_t> ones(size, 1), rc(size, 0);
  a2a_host(scnt.data(), ones, rc.data(), ones, (int)sizeof(int64_t));
  return rc;
}

int64_t Comm::allsum(int64_t v) const {
  if (!distributed()) return v;
  std::vector<int64_t> s(size, v);
  auto r = exchange_counts(s);
  int64_t t = 0;
  for (int64_t x : r) t += x;
  return t;
}

void Comm::allsum(std::vector<double>& v) const {
  if (!distributed() || v.empty()) return;
  std::vector<double> all = allgather(v);
  const size_t m = v.size();
  for (size_t i = 0; i < m; i++) { double s = 0.0; for (int q = 0; q < size; q++) s += all[(size_t)q * m + i]; v[i] = s; }
}

void Comm::release() {
  if (release_fn && ctx) { release_fn(ctx); ctx = nullptr; alltoallv = nullptr; alloc = nullptr; }
  release_fn = nullptr; native = false;
  sarena_ = rarena_ = nullptr; scap_ = rcap_ = 0;
}

static double* grow(const Comm& c, double*& arena, int64_t& cap, int64_t need) {
  if (need <= cap && arena) return arena;
  HYMLS_CHECK(c.alloc != nullptr, -2, "sharded run without an arena allocator (hymls_mi_set_comm)");
  cap = std::max<int64_t>(need + need / 4, 1024);
  arena = (double*)c.alloc(c.ctx, cap * (int64_t)sizeof(double));
  HYMLS_CHECK(arena != nullptr, -3, "exchange arena allocation failed in the transport callback");
  return arena;
}
double* Comm::send_arena(int64_t n) const { return grow(*this, sarena_, scap_, n); }
double* Comm::recv_arena(int64_t n) const { return grow(*this, rarena_, rcap_, n); }

Exchange::~Exchange() { dev::free(d_sidx); dev::free(d_ridx); }

void Exchange::build(const Comm& c, const std::vector<std::vector<int64_t>>& want_keys, const std::vector<ivec>& want_dst,
                     const std::function<int32_t(int64_t)>& resolve) {
  comm = &c;
  dev::free(d_sidx); dev::free(d_ridx);
  d_sidx = d_ridx = nullptr;
  h_sidx.clear(); h_ridx.clear();
  scnt.assign(c.size, 0); rcnt.assign(c.size, 0);
  auto asked = c.exchange_lists(want_keys);   // asked[q]: keys rank q wants from me
  for (int q = 0; q < c.size; q++) {
    rcnt[q] = (int64_t)want_keys[q].size();
    HYMLS_CHECK(want_dst[q].size() == want_keys[q].size(), -3, "exchange plan: inconsistent request lists");
    h_ridx.insert(h_ridx.end(), want_dst[q].begin(), want_dst[q].end());
    scnt[q] = (int64_t)asked[q].size();
    for (int64_t k : asked[q]) {
      const int32_t s = resolve(k);
      HYMLS_CHECK(s >= 0, -3, "exchange plan: rank " + std::to_string(q) + " asked rank " + std::to_string(c.rank) +
                                  " for entry " + std::to_string(k) + " which it does not hold");
      h_sidx.push_back(s);
    }
  }
  nsend = (int64_t)h_sidx.size(); nrecv = (int64_t)h_ridx.size();
  any = c.allsum(nsend) > 0;
  if (nsend) d_sidx = dev::upload(h_sidx);
  if (nrecv) d_ridx = dev::upload(h_ridx);
  if (any) { c.send_arena(std::max(nsend, nrecv)); c.recv_arena(std::max(nsend, nrecv)); }
}

void Exchange::forward(const double* src, double* dst) const {
  if (!any) return;
  double* sb = comm->send_arena(nsend);
  double* rb = comm->recv_arena(nrecv);
  if (nsend) dev::gather(nsend, d_sidx, src, sb);
  const int ierr = comm->alltoallv(comm->ctx, sb, scnt.data(), rb, rcnt.data(), (int32_t)sizeof(double), 1);
  HYMLS_CHECK(ierr == 0, -3, std::string("device all-to-all failed in the transport ") + rccl_last_error(*comm));
  if (nrecv) dev::scatter(nrecv, d_ridx, rb, dst);
}

void Exchange::backward(const double* src, double* dst, bool add) const {
  if (!any) return;
  double* sb = comm->send_arena(nrecv);
  double* rb = comm->recv_arena(nsend);
  if (nrecv) dev::gather(nrecv, d_ridx, src, sb);
  const int ierr = comm->alltoallv(comm->ctx, sb, rcnt.data(), rb, scnt.data(), (int32_t)sizeof(double), 1);
  HYMLS_CHECK(ierr == 0, -3, std::string("device all-to-all failed in the transport ") + rccl_last_error(*comm));
  if (!nsend) return;
  if (!add) { dev::scatter(nsend, d_sidx, rb, dst); return; }
  // peer by peer (a position may get contributions from several peers): fixed order of the additions
  int64_t off = 0;
  for (int q = 0; q < comm->size; q++) { dev::scatter_add(scnt[q], d_sidx + off, rb + off, dst); off += scnt[q]; }
}

}  // namespace hymls
