// engine.hip -- core of the graph engine: parameters, arenas, op emitters, execution (see engine.h).
#include "engine.h"

#include <cstdio>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

namespace sdod {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void check_rc(int rc) {
    if (rc != 0) throw Error(rc, get_last_error());
}

template <typename F>
static void parallel_for(int64_t n, F&& fn) {
    int nt = (int)std::min<int64_t>(n, std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t b = t * per, e = std::min(n, b + per);
        if (b < e) th.emplace_back([=, &fn]() { fn(b, e); });
    }
    for (auto& t : th) t.join();
}

// ------------------------------------------------------------------------------------------ lifecycle
Graph::Graph(int kind, const sdod_model_config& cfg, int batch) : kind_(kind), cfg_(cfg), batch_(batch) {
    SDOD_REQUIRE(kind >= SDOD_GRAPH_UNET && kind <= SDOD_GRAPH_TEMB, "unknown graph kind");
    SDOD_REQUIRE(batch > 0 && batch <= 64, "batch must be in [1, 64]");
    SDOD_REQUIRE(cfg.model_channels > 0 && cfg.model_channels % 64 == 0, "model_channels must be a multiple of 64");
    SDOD_REQUIRE(cfg.context_dim % 64 == 0, "context_dim must be a multiple of 64");
    mode_ = DECLARE;
    arena_reset();
    build();
    // the parameter table is complete and readable without a device (checkpoint conversion runs on CPU-only hosts);
    // device memory is claimed by the first set_param / load_file / finalize
}

Graph::~Graph() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (capture_stream_) (void)hipStreamDestroy(capture_stream_);
    if (side_stream_) (void)hipStreamDestroy(side_stream_);
    (void)hipFree(fix_counters_);
    for (hipEvent_t e : pf_events_) (void)hipEventDestroy(e);
    if (hip_graph_) (void)hipGraphDestroy(hip_graph_);
    for (auto& s : inputs_) (void)hipFree(s.ptr);
    for (auto& s : outputs_) (void)hipFree(s.ptr);
    (void)hipFree(weight_base_);
    (void)hipFree(qscale_);
    (void)hipFree(qoff_);
    (void)hipFree(arena_base_);
    (void)hipFree(ws_);
    (void)hipFree(gn_ws_);
    (void)hipFree(kv_all_);
    for (void* d : derived_) (void)hipFree(d);
}

// ------------------------------------------------------------------------------------------ parameters
static size_t param_dev_bytes(const Param& p) {
    const auto& s = p.shape;
    if (p.quant) return (size_t)s[0] * s[1] * (p.kind == PK_CONV3 ? 9 : 1); // one byte per code
    switch (p.kind) {
    case PK_CONV3: return (size_t)s[0] * s[1] * 9 * sizeof(f16);
    case PK_CONV3_SMALL: return (size_t)s[0] * 64 * sizeof(f16);
    case PK_CONV1:
    case PK_LINEAR:
    case PK_LINEAR_GEGLU:
    case PK_EMBED: return (size_t)s[0] * s[1] * sizeof(f16);
    case PK_VEC:
    case PK_VEC_GEGLU: return (size_t)s[0] * sizeof(float);
    case PK_MAT_F32: {
        size_t n = 1;
        for (auto d : s) n *= (size_t)d;
        return n * sizeof(float);
    }
    }
    return 0;
}

int Graph::P(const std::string& name, std::vector<int64_t> shape, ParamKind kind, const std::string& group) {
    if (mode_ != DECLARE) {
        auto it = pindex_.find(name);
        if (it == pindex_.end()) throw Error(INTERNAL_ERROR, "parameter not declared: " + name);
        return it->second;
    }
    if (pindex_.count(name)) throw Error(INTERNAL_ERROR, "duplicate parameter: " + name);
    Param p;
    p.name = name;
    p.shape = std::move(shape);
    p.kind = kind;
    p.quant = quant_decl_ && (kind == PK_CONV3 || kind == PK_CONV1 || kind == PK_LINEAR || kind == PK_LINEAR_GEGLU);
    p.dev_bytes = param_dev_bytes(p);
    if (!group.empty()) {
        auto it = gindex_.find(group);
        if (it == gindex_.end()) {
            gindex_[group] = (int)groups_.size();
            groups_.emplace_back();
            it = gindex_.find(group);
        }
        p.group = it->second;
        groups_[p.group].push_back((int)params_.size());
    }
    pindex_[name] = (int)params_.size();
    params_.push_back(std::move(p));
    return (int)params_.size() - 1;
}

int Graph::Pc(const std::string& name, std::vector<int64_t> shape, ParamKind kind, int ld, int col_off, int owner) {
    const int idx = P(name, std::move(shape), kind);
    if (mode_ == DECLARE) {
        Param& p = params_[idx];
        p.ld = ld;
        p.col_off = col_off;
        p.owner = owner;
        p.dev_bytes = owner < 0 ? (size_t)p.shape[0] * ld * sizeof(f16) : 0;
    }
    return idx;
}

void Graph::allocate_weights() {
    if (weight_base_) return;
    // groups first (members contiguous, declaration order), then everything else; 256-byte alignment per block
    size_t off = 0;
    std::vector<size_t> offs(params_.size(), 0);
    for (auto& g : groups_) {
        off = align_up(off, 256);
        for (int idx : g) {
            if (params_[idx].dev_bytes % 16) throw Error(INTERNAL_ERROR, "grouped parameter size must be a multiple of 16: " + params_[idx].name);
            offs[idx] = off;
            off += params_[idx].dev_bytes;
        }
    }
    for (size_t i = 0; i < params_.size(); ++i) {
        if (params_[i].group >= 0) continue;
        off = align_up(off, 256);
        offs[i] = off;
        off += params_[i].dev_bytes;
    }
    weight_bytes_ = align_up(off, 256);
    SDOD_HIP_CHECK(hipMalloc((void**)&weight_base_, weight_bytes_));
    SDOD_HIP_CHECK(hipMemset(weight_base_, 0, weight_bytes_));
    for (size_t i = 0; i < params_.size(); ++i) params_[i].dev = weight_base_ + offs[i];
    // rows of the uint8 weights, numbered in arena order (ascending device offset): members of a fused group are contiguous
    {
        std::vector<size_t> order;
        for (size_t i = 0; i < params_.size(); ++i)
            if (params_[i].quant) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return offs[a] < offs[b]; });
        qrows_ = 0;
        for (size_t i : order) {
            params_[i].qrow = qrows_;
            qrows_ += (size_t)params_[i].shape[0];
        }
        if (qrows_) {
            SDOD_HIP_CHECK(hipMalloc((void**)&qscale_, qrows_ * sizeof(float)));
            SDOD_HIP_CHECK(hipMalloc((void**)&qoff_, qrows_ * sizeof(float)));
        }
    }
    for (auto& p : params_)
        if (p.owner >= 0) p.dev = params_[p.owner].dev + (size_t)p.col_off * sizeof(f16);
}

template <typename S>
static void pack_param_host(const Param& p, const S* src, char* dst_raw) {
    const auto& s = p.shape;
    switch (p.kind) {
    case PK_CONV3:
    case PK_CONV3_SMALL: {
        const int64_t co = s[0], ci = s[1];
        const int64_t kd = p.kind == PK_CONV3 ? 9 * ci : 64;
        f16* dst = reinterpret_cast<f16*>(dst_raw);
        parallel_for(co, [&](int64_t b, int64_t e) {
            for (int64_t o = b; o < e; ++o) {
                f16* drow = dst + o * kd;
                if (p.kind == PK_CONV3_SMALL) std::memset(drow, 0, kd * sizeof(f16));
                const S* srow = src + o * ci * 9;
                for (int64_t c = 0; c < ci; ++c)
                    for (int t = 0; t < 9; ++t) drow[t * ci + c] = (f16)(float)srow[c * 9 + t];
            }
        });
        break;
    }
    case PK_CONV1:
    case PK_LINEAR:
    case PK_EMBED: {
        const int64_t n = s[0] * s[1];
        f16* dst = reinterpret_cast<f16*>(dst_raw);
        parallel_for(n, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) dst[i] = (f16)(float)src[i];
        });
        break;
    }
    case PK_LINEAR_GEGLU: { // value row j -> (j/16)*32 + j%16 ; gate row H+j -> (j/16)*32 + 16 + j%16
        const int64_t rows = s[0], k = s[1], H = rows / 2;
        f16* dst = reinterpret_cast<f16*>(dst_raw);
        parallel_for(rows, [&](int64_t b, int64_t e) {
            for (int64_t r = b; r < e; ++r) {
                const int64_t j = r < H ? r : r - H;
                const int64_t nr = (j / 16) * 32 + (r < H ? 0 : 16) + j % 16;
                for (int64_t c = 0; c < k; ++c) dst[nr * k + c] = (f16)(float)src[r * k + c];
            }
        });
        break;
    }
    case PK_VEC_GEGLU: {
        const int64_t rows = s[0], H = rows / 2;
        float* dst = reinterpret_cast<float*>(dst_raw);
        for (int64_t r = 0; r < rows; ++r) {
            const int64_t j = r < H ? r : r - H;
            dst[(j / 16) * 32 + (r < H ? 0 : 16) + j % 16] = (float)src[r];
        }
        break;
    }
    case PK_VEC:
    case PK_MAT_F32: {
        int64_t n = 1;
        for (auto d : s) n *= d;
        float* dst = reinterpret_cast<float*>(dst_raw);
        for (int64_t i = 0; i < n; ++i) dst[i] = (float)src[i];
        break;
    }
    }
}

void Graph::set_param(const std::string& name, const void* data, int dtype, const int64_t* shape, int ndim) {
    allocate_weights();
    SDOD_REQUIRE(!finalized_, "parameters cannot change after finalize() (weights are repacked / folded in place)");
    auto it = pindex_.find(name);
    SDOD_REQUIRE(it != pindex_.end(), "unknown parameter '" + name + "'");
    SDOD_REQUIRE(data != nullptr, "null data for '" + name + "'");
    SDOD_REQUIRE(dtype == SDOD_F32 || dtype == SDOD_F16 || dtype == SDOD_U8Q, "dtype must be SDOD_F32, SDOD_F16 or SDOD_U8Q");
    Param& p = params_[it->second];
    int64_t want = 1, got = 1;
    for (auto d : p.shape) want *= d;
    for (int i = 0; i < ndim; ++i) got *= shape[i];
    // int8-weight checkpoints (BASELINE config 5, "mirrors the QNN quant path"): dequantised here, once, with the reference's
    // own arithmetic -- real = (q + offset) * scale evaluated in double, then rounded to float (qnn_context.cpp:1018-1033);
    // the kernels then run on the fp16 image of those values, exactly as they do for an fp16 checkpoint
    if (p.quant) {
        // the tensor stays affine uint8: repack the CODES exactly as the fp16 path repacks values (KRSC for 3x3 convs, 16-row
        // value / gate interleave for GEGLU) and record its encoding for every output row
        SDOD_REQUIRE(dtype == SDOD_U8Q, "graph was created with weight_quant != 0: '" + name + "' must be given as SDOD_U8Q");
        bool same = (int)p.shape.size() == ndim;
        for (int i = 0; same && i < ndim; ++i) same = p.shape[i] == shape[i];
        if (!same && want == got && (p.kind == PK_CONV1 || p.kind == PK_LINEAR) && ndim >= 2 && shape[0] == p.shape[0]) same = true;
        SDOD_REQUIRE(same, "shape mismatch for '" + name + "'");
        float scale;
        int32_t offset;
        std::memcpy(&scale, data, 4);
        std::memcpy(&offset, static_cast<const char*>(data) + 4, 4);
        // the kernels fold the zero point into the fragment expansion: q + offset must be an exact fp16 integer (gemm.hip: u8x8_to_f16)
        SDOD_REQUIRE(offset >= -1024 && offset <= 0, "'" + name + "': uint8 encoding offset must be in [-1024, 0] (QNN: -zero_point of a uint8 tensor)");
        const uint8_t* q = static_cast<const uint8_t*>(data) + 8;
        const int64_t rows = p.shape[0], kd = got / rows;
        std::vector<uint8_t> staging((size_t)got);
        uint8_t* dst = staging.data();
        if (p.kind == PK_CONV3) {
            const int64_t ci = p.shape[1];
            parallel_for(rows, [&](int64_t b, int64_t e) {
                for (int64_t o = b; o < e; ++o)
                    for (int64_t c = 0; c < ci; ++c)
                        for (int t = 0; t < 9; ++t) dst[o * kd + t * ci + c] = q[(o * ci + c) * 9 + t];
            });
        } else if (p.kind == PK_LINEAR_GEGLU) {
            const int64_t H = rows / 2;
            parallel_for(rows, [&](int64_t b, int64_t e) {
                for (int64_t r = b; r < e; ++r) {
                    const int64_t j = r < H ? r : r - H;
                    const int64_t nr = (j / 16) * 32 + (r < H ? 0 : 16) + j % 16;
                    std::memcpy(dst + nr * kd, q + r * kd, (size_t)kd);
                }
            });
        } else {
            std::memcpy(dst, q, (size_t)got);
        }
        SDOD_REQUIRE(p.ld == 0, "column-block parameters are not used with uint8 weights");
        SDOD_HIP_CHECK(hipMemcpy(p.dev, staging.data(), (size_t)got, hipMemcpyHostToDevice));
        std::vector<float> sv((size_t)rows, scale), ov((size_t)rows, (float)(offset + 128));
        SDOD_HIP_CHECK(hipMemcpy(qscale_ + p.qrow, sv.data(), sv.size() * sizeof(float), hipMemcpyHostToDevice));
        SDOD_HIP_CHECK(hipMemcpy(qoff_ + p.qrow, ov.data(), ov.size() * sizeof(float), hipMemcpyHostToDevice));
        p.set = true;
        return;
    }
    std::vector<float> deq;
    if (dtype == SDOD_U8Q) {
        float scale;
        int32_t offset;
        std::memcpy(&scale, data, 4);
        std::memcpy(&offset, static_cast<const char*>(data) + 4, 4);
        const uint8_t* q = static_cast<const uint8_t*>(data) + 8;
        deq.resize((size_t)got);
        float* dst = deq.data();
        parallel_for(got, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) dst[i] = (float)((double)((int32_t)q[i] + offset) * (double)scale);
        });
        data = deq.data();
        dtype = SDOD_F32;
    }
    bool same = (int)p.shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = p.shape[i] == shape[i];
    // a [Cout][Cin] matrix is accepted for a 1x1 conv and vice versa (SD2.x stores proj_in/out as Linear)
    if (!same && want == got && (p.kind == PK_CONV1 || p.kind == PK_LINEAR) && ndim >= 2 && shape[0] == p.shape[0]) same = true;
    if (!same) {
        std::string m = "shape mismatch for '" + name + "': expected [";
        for (auto d : p.shape) m += std::to_string(d) + ",";
        m += "] got [";
        for (int i = 0; i < ndim; ++i) m += std::to_string(shape[i]) + ",";
        throw Error(INVALID_ARGUMENT, m + "]");
    }
    Param dense = p; // packing always produces the dense image; strided parameters are placed with a 2-D copy
    dense.ld = 0;
    const size_t dense_bytes = param_dev_bytes(dense);
    std::vector<char> staging(dense_bytes);
    if (dtype == SDOD_F32) pack_param_host(dense, reinterpret_cast<const float*>(data), staging.data());
    else pack_param_host(dense, reinterpret_cast<const f16*>(data), staging.data());
    if (p.ld > 0) {
        const size_t row_bytes = dense_bytes / (size_t)p.shape[0];
        SDOD_HIP_CHECK(hipMemcpy2D(p.dev, (size_t)p.ld * sizeof(f16), staging.data(), row_bytes, row_bytes, (size_t)p.shape[0],
                                   hipMemcpyHostToDevice));
    } else {
        SDOD_HIP_CHECK(hipMemcpy(p.dev, staging.data(), dense_bytes, hipMemcpyHostToDevice));
    }
    p.set = true;
}

// .sdodw container: "SDODW001", u64 count, then per tensor {u32 name_len, name, u32 dtype, u32 ndim, u64 dims[ndim],
// u64 offset, u64 nbytes}; payloads at absolute file offsets.  Written by sdod.amd.weights.save().
void Graph::load_file(const std::string& path, const std::string& prefix) {
    // (device memory is claimed by the first set_param: a malformed container is rejected before any device call)
    int fd = ::open(path.c_str(), O_RDONLY);
    SDOD_REQUIRE(fd >= 0, "cannot open weight file " + path);
    struct stat stt;
    if (fstat(fd, &stt) != 0) {
        ::close(fd);
        throw Error(INVALID_ARGUMENT, "cannot stat " + path);
    }
    const size_t fsize = (size_t)stt.st_size;
    void* map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    SDOD_REQUIRE(map != MAP_FAILED, "mmap failed for " + path);
    struct Unmap {
        void* p;
        size_t n;
        ~Unmap() { munmap(p, n); }
    } guard{map, fsize};
    const char* base = static_cast<const char*>(map);
    size_t pos = 0;
    auto need = [&](size_t n) { SDOD_REQUIRE(pos + n <= fsize, "truncated weight file " + path); };
    need(16);
    SDOD_REQUIRE(std::memcmp(base, "SDODW001", 8) == 0, "bad magic in " + path);
    uint64_t count;
    std::memcpy(&count, base + 8, 8);
    pos = 16;
    for (uint64_t t = 0; t < count; ++t) {
        uint32_t nl, dt, nd;
        need(4); std::memcpy(&nl, base + pos, 4); pos += 4;
        need(nl); std::string name(base + pos, nl); pos += nl;
        need(8); std::memcpy(&dt, base + pos, 4); std::memcpy(&nd, base + pos + 4, 4); pos += 8;
        SDOD_REQUIRE(nd <= 8, "bad ndim in " + path);
        int64_t dims[8];
        need(8 * nd + 16);
        for (uint32_t i = 0; i < nd; ++i) { uint64_t d; std::memcpy(&d, base + pos, 8); pos += 8; dims[i] = (int64_t)d; }
        uint64_t off, nb;
        std::memcpy(&off, base + pos, 8); std::memcpy(&nb, base + pos + 8, 8); pos += 16;
        SDOD_REQUIRE(off <= fsize && nb <= fsize - off, "tensor payload out of range in " + path); // no u64 wrap
        SDOD_REQUIRE(dt == SDOD_F16 || dt == SDOD_F32 || dt == SDOD_U8Q, "bad dtype for '" + name + "' in " + path);
        // the payload must hold exactly prod(dims) elements (+ the {scale, offset} prefix of an affine-uint8 tensor):
        // set_param reads that many bytes from the mapping
        uint64_t numel = 1;
        for (uint32_t i = 0; i < nd; ++i) {
            SDOD_REQUIRE(dims[i] > 0 && (uint64_t)dims[i] <= (uint64_t(1) << 40) / numel, "bad dims for '" + name + "' in " + path);
            numel *= (uint64_t)dims[i];
        }
        const uint64_t want = dt == SDOD_U8Q ? 8 + numel : numel * (dt == SDOD_F16 ? 2 : 4);
        SDOD_REQUIRE(nb == want, "payload size of '" + name + "' does not match its dims in " + path);
        if (name.compare(0, prefix.size(), prefix) != 0) continue;
        const std::string local = name.substr(prefix.size());
        if (!pindex_.count(local)) continue;
        set_param(local, base + off, (int)dt, dims, (int)nd);
    }
}

// ------------------------------------------------------------------------------------------ arenas
static char* const kFakeBase = reinterpret_cast<char*>(uintptr_t(1) << 40);

void Graph::arena_reset() {
    free_.clear();
    used_.clear();
    if (mode_ == REAL) {
        free_[0] = arena_cap_;
    } else {
        free_[0] = size_t(1) << 39;
        arena_high_ = 0;
    }
}

f16* Graph::alloc(size_t halves) {
    const size_t bytes = align_up(std::max<size_t>(halves * sizeof(f16), 256), 256);
    for (auto it = free_.begin(); it != free_.end(); ++it) {
        if (it->second >= bytes) {
            const size_t off = it->first, sz = it->second;
            free_.erase(it);
            if (sz > bytes) free_[off + bytes] = sz - bytes;
            used_[off] = bytes;
            arena_high_ = std::max(arena_high_, off + bytes);
            char* base = mode_ == REAL ? arena_base_ : kFakeBase;
            return reinterpret_cast<f16*>(base + off);
        }
    }
    throw Error(INTERNAL_ERROR, "activation arena exhausted");
}

void Graph::release(const void* p) {
    if (!p) return;
    if (pending_.active && pending_.d.residual == p) flush_pending(); // the deferred reduce still reads this tensor
    char* base = mode_ == REAL ? arena_base_ : kFakeBase;
    const size_t off = (size_t)(reinterpret_cast<const char*>(p) - base);
    auto it = used_.find(off);
    if (it == used_.end()) throw Error(INTERNAL_ERROR, "arena release of unknown block");
    size_t sz = it->second;
    used_.erase(it);
    size_t start = off;
    auto nxt = free_.lower_bound(off);
    if (nxt != free_.end() && nxt->first == off + sz) {
        sz += nxt->second;
        nxt = free_.erase(nxt);
    }
    if (nxt != free_.begin()) {
        auto prv = std::prev(nxt);
        if (prv->first + prv->second == off) {
            start = prv->first;
            sz += prv->second;
            free_.erase(prv);
        }
    }
    free_[start] = sz;
}

Act Graph::act(int n, int h, int w, int c) {
    Act a;
    a.n = n; a.h = h; a.w = w; a.c = c;
    a.p = alloc(a.numel());
    return a;
}

void* Graph::io_alloc(std::vector<IoSlot>& v, size_t bytes) {
    IoSlot s;
    s.bytes = bytes;
    if (mode_ == REAL) {
        SDOD_HIP_CHECK(hipMalloc(&s.ptr, align_up(bytes, 256)));
        SDOD_HIP_CHECK(hipMemset(s.ptr, 0, align_up(bytes, 256)));
        v.push_back(s);
        return s.ptr;
    }
    return kFakeBase + (size_t(1) << 39) + v.size() * 4096; // never dereferenced
}

IoSlot Graph::io(bool output, int index) const {
    const auto& v = output ? outputs_ : inputs_;
    SDOD_REQUIRE(finalized_, "graph not finalized");
    SDOD_REQUIRE(index >= 0 && index < (int)v.size(), "io index out of range");
    return v[index];
}

// ------------------------------------------------------------------------------------------ emitters
// ---- tile autotuning: "measure, don't guess".  Every distinct GEMM shape of a graph is timed once per process with
// each candidate tile configuration of gemm.hip on the real operands and the fastest is baked into the launch list.
// The timing is COLD-CACHE (sdod_gemm_time_cold: weights evicted to HBM, activations re-touched) because that is what a
// launch meets inside a replay: the UNet's weights are 1.7 GB, the Infinity Cache 256 MiB.  SDOD_AUTOTUNE=0 disables the
// tuner (gemm.hip's static heuristic decides), SDOD_AUTOTUNE=hot ranks with back-to-back launches instead.
namespace {
constexpr size_t kPrefetchMinBytes = (size_t)12 << 20; // weight matrices at least this big are prefetched (Graph::run_ops)
const int kCandidates[] = {1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, // (25, 26 spill in their epilogue only; the tuner decides)
                           37, 38, 39, 40, 41, 42, 43, 44, 45, // halo-patch convolution tiles: rejected by every other descriptor
                           46, 47, 48,
                           49, 50, 51, 52, // halo-patch tiles of 96 / 192 rows (image rows that are multiples of 3: config 5)
                           53, 54, 55,     // A-panel tiles (short-K wide-N Linears): rejected by every other descriptor
                           56, 57, 58, 59, 60, 61}; // ring tiles whose waves own a head's 80 columns: the softmax-epilogue GEMM of the folded cross-attention

struct ShapeKey {
    static constexpr int kFields = 14;
    int v[kFields];
    bool operator<(const ShapeKey& o) const { return std::lexicographical_compare(v, v + kFields, o.v, o.v + kFields); }
};
// Picks come from two text files (one shape per line: the 14 integers of the shape key, then tile + 1000 * split_k; lines of
// the round-1/2 format -- 12 key integers, no w_in / n_img -- are still read: their convolutions were all square):
//   * the SHIPPED table tune/gfx950.tune next to lib/ (located through dladdr): the picks for the headline shapes, made on
//     an MI355X by tools/make_tune_cache.py and committed, so that every process -- bench, rocprofv3 passes, the C API, a
//     service restart -- builds the SAME launch list without timing anything, and images are bit-identical across processes
//     (SDOD_TUNE_DEFAULT=0 ignores it);
//   * SDOD_TUNE_CACHE=<file>: read after (so it overrides) the shipped table; shapes found in neither are timed once and
//     appended to it.  Picks depend on the build and the device: regenerate the table when the tiles change.
const char* tune_cache_path() {
    const char* e = std::getenv("SDOD_TUNE_CACHE");
    return (e && e[0]) ? e : nullptr;
}
std::string shipped_tune_path() {
    const char* e = std::getenv("SDOD_TUNE_DEFAULT");
    if (e && e[0] == '0') return "";
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&sdod_model_config_sd14), &info) || !info.dli_fname) return "";
    std::string lib = info.dli_fname; // .../stable-diffusion-on-device_amd/lib/libsdod.so
    const size_t a = lib.rfind('/');
    if (a == std::string::npos) return "";
    const size_t b = lib.rfind('/', a - 1);
    if (b == std::string::npos) return "";
    return lib.substr(0, b) + "/tune/gfx950.tune";
}
struct TuneEntry {
    int v;          // tile + 1000 * split_k
    bool from_file; // read from a table (as opposed to timed by this process)
};
struct TuneCache { // process-wide; graphs may be built from several threads (libsdod_setup loads its models in parallel in the reference)
    std::mutex mu;
    std::map<ShapeKey, TuneEntry> map;
    bool loaded = false;
};
void tune_cache_read(const char* path, std::map<ShapeKey, TuneEntry>& c) {
    FILE* f = std::fopen(path, "r");
    if (!f) return;
    char line[512];
    while (std::fgets(line, sizeof line, f)) {
        long val[ShapeKey::kFields + 2];
        int n = 0;
        char* cur = line;
        while (n < ShapeKey::kFields + 2) {
            char* endp = nullptr;
            const long x = std::strtol(cur, &endp, 10);
            if (endp == cur) break;
            val[n++] = x;
            cur = endp;
        }
        ShapeKey k{};
        int v;
        if (n == ShapeKey::kFields + 1) {
            for (int i = 0; i < ShapeKey::kFields; ++i) k.v[i] = (int)val[i];
            v = (int)val[ShapeKey::kFields];
        } else if (n == 13) { // round-1/2 line: {a_mode, M, N, K, c0, c1, stride, upsample, ksize, h_in, flags, lda}: square images
            for (int i = 0; i < 12; ++i) k.v[i] = (int)val[i];
            k.v[12] = k.v[13] = 0;
            if (k.v[0] == SDOD_A_CONV3X3 && k.v[9] > 0) {
                const int ups = k.v[7] ? 1 : 0, stride = k.v[6] > 0 ? k.v[6] : 1;
                const int ho = (k.v[9] << ups) / stride;
                k.v[12] = k.v[9];                              // w_in = h_in
                k.v[13] = ho > 0 ? k.v[1] / (ho * ho) : 0;     // n_img = M / (h_out * w_out)
            }
            v = (int)val[12];
        } else {
            continue;
        }
        const int tile = v % 1000, split = v / 1000;
        if (tile >= 1 && tile <= sdod_gemm_num_tiles() && split >= 1 && split <= 64) c[k] = TuneEntry{v, true}; // a stale table is re-tuned, not trusted
    }
    std::fclose(f);
}
TuneCache& tune_cache() {
    static TuneCache c;
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.loaded) {
        c.loaded = true;
        const std::string shipped = shipped_tune_path();
        if (!shipped.empty()) tune_cache_read(shipped.c_str(), c.map);
        if (const char* path = tune_cache_path()) tune_cache_read(path, c.map);
    }
    return c;
}
void tune_cache_append(const ShapeKey& k, int v) {
    const char* path = tune_cache_path();
    if (!path) return;
    if (FILE* f = std::fopen(path, "a")) {
        for (int i = 0; i < ShapeKey::kFields; ++i) std::fprintf(f, "%d ", k.v[i]);
        std::fprintf(f, "%d\n", v);
        std::fclose(f);
    }
}
bool autotune_enabled() {
    const char* e = std::getenv("SDOD_AUTOTUNE");
    return !(e && e[0] == '0');
}
bool autotune_cold() {
    const char* e = std::getenv("SDOD_AUTOTUNE");
    return !(e && e[0] == 'h');
}
constexpr size_t kSweepBytes = (size_t)384 << 20; // > 8 x 4 MiB L2 + 256 MiB Infinity Cache
// scratch the cold timing sweeps; allocated on first use, released by release_tune_scratch() at the end of finalize()
void*& tune_scratch() {
    static void* p = nullptr;
    return p;
}
ShapeKey key_of(const sdod_gemm_desc& d) {
    // w_in and n_img are part of the key: which halo-patch tiles take a convolution depends on both (halo_geometry), so two
    // convolutions with equal M and h_in but different (n_img, w_in) must not share a pick
    return ShapeKey{{d.a_mode, d.M, d.N, d.K, d.c0, d.c1, d.stride, d.upsample, d.ksize, d.h_in,
                     (d.residual ? 1 : 0) + (d.geglu ? 2 : 0) + 4 * d.tc0 + 16384 * d.tc1 + (d.ln ? (1 << 30) : 0) + (d.wq ? (1 << 29) : 0) + (d.softmax_cols ? (1 << 28) : 0) + (d.w_img_stride ? (1 << 27) : 0), d.lda,
                     d.a_mode == SDOD_A_CONV3X3 ? d.w_in : 0, d.a_mode == SDOD_A_CONV3X3 ? d.n_img : 0}};
}
// a table pick the library would refuse at launch (a halo-patch / A-panel tile that does not take this descriptor: a stale or
// hand-edited table) is not used
bool pick_runs(const sdod_gemm_desc& d, int v) {
    sdod_gemm_desc c = d;
    c.tile = v % 1000;
    c.split_k = v / 1000;
    int info[7] = {0};
    if (sdod_gemm_tile_info(c.tile, info) != 0) return false;
    if (info[5] == 2) return sdod_gemm_halo_ok(&c, c.tile) != 0;
    if (info[5] == 3) return sdod_gemm_panel_ok(&c, c.tile) != 0;
    return true;
}
} // namespace

void Graph::flush_pending() {
    if (!pending_.active) return;
    pending_.active = false;
    const sdod_gemm_desc d2 = pending_.d;
    sink().push_back(Op{[d2](hipStream_t st) { check_rc(sdod_gemm_f16(&d2, st)); }, "splitk_reduce", 0.0, pending_.bytes, pending_.detail});
}

void Graph::emit_gemm(sdod_gemm_desc d) {
    settle();
    if (mode_ == DECLARE) return;
    // (N <= 16: the skinny default tile -- except the UNet's output convolution, 320 -> 4 channels, where a halo-patch tile without
    // split-K beats it by 7 us and a reduce launch: tools/convout_bench.py)
    const bool tune = autotune_enabled() && (d.N > 16 || (d.a_mode == SDOD_A_CONV3X3 && d.ksize != 1 && d.N % 4 == 0));
    if (tune) {
        for (int t : kCandidates) {
            sdod_gemm_desc c = d;
            c.tile = t;
            c.split_k = 0;
            int tt = 0, sp = 1;
            (void)sdod_gemm_plan(&c, &tt, &sp);
            c.split_k = sp > 1 ? std::min(64, sp * 2) : 1;
            if (c.split_k > 1 && d.N % 4 == 0) ws_need_ = std::max(ws_need_, sdod_gemm_workspace_bytes(&c));
        }
    }
    ws_need_ = std::max(ws_need_, sdod_gemm_workspace_bytes(&d));
    if (mode_ != REAL) return;
    if (tune) {
        d.workspace = ws_;
        d.workspace_bytes = ws_bytes_;
        d.fix_counters = std::getenv("SDOD_GEMM_FIXUP") && std::getenv("SDOD_GEMM_FIXUP")[0] == '1' ? fix_counters_ : nullptr;
        const ShapeKey key = key_of(d);
        TuneCache& tc = tune_cache();
        int pick = 0;
        bool from_file = false;
        {
            std::lock_guard<std::mutex> lk(tc.mu);
            auto it = tc.map.find(key);
            if (it != tc.map.end() && pick_runs(d, it->second.v)) {
                pick = it->second.v;
                from_file = it->second.from_file;
            }
        }
        if (pick && from_file) ++tune_hits_;
        else ++tune_misses_; // timed by this process (now, or earlier by another graph): the launch list is not the table's
        if (!pick) {
            int best = 0;
            float best_ms = 1e30f;
            // candidates: every tile x {heuristic split-K, half, double, none (one launch instead of two)}
            const int KT = d.K / 64;
            for (int t : kCandidates) {
                sdod_gemm_desc h = d;
                h.tile = t;
                h.split_k = 0;
                int tt = 0, sp = 1;
                (void)sdod_gemm_plan(&h, &tt, &sp);
                if (tt != t) continue; // the plan runs this descriptor on another tile (a fusion `t` does not carry): that tile is a candidate itself
                int tried[4] = {sp, 1, sp / 2, sp > 1 ? sp * 2 : 0};
                for (int k = 0; k < 4; ++k) {
                    const int want = tried[k];
                    bool dup = want < 1 || want > 64 || (want > 1 && KT / want < 2) || (want > 1 && d.N % 4 != 0);
                    for (int j = 0; j < k && !dup; ++j) dup = tried[j] == want;
                    if (dup) continue;
                    sdod_gemm_desc c = d;
                    c.tile = t;
                    c.split_k = want;
                    if (sdod_gemm_workspace_bytes(&c) > ws_bytes_) continue;
                    float ms = 0.f;
                    if (autotune_cold()) {
                        if (!tune_scratch()) {
                            SDOD_HIP_CHECK(hipMalloc(&tune_scratch(), kSweepBytes));
                            SDOD_HIP_CHECK(hipMemset(tune_scratch(), 0, kSweepBytes));
                        }
                        float avg = 0.f;
                        if (sdod_gemm_time_cold(&c, nullptr, 4, tune_scratch(), kSweepBytes, &avg, &ms) != 0) continue;
                    } else if (sdod_gemm_time(&c, nullptr, 4, &ms) != 0) {
                        continue;
                    }
                    if (ms < best_ms) {
                        best_ms = ms;
                        best = t + 1000 * want;
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(tc.mu);
                tc.map[key] = TuneEntry{best, false};
                tune_cache_append(key, best);
            }
            pick = best;
        }
        d.tile = pick % 1000;
        d.split_k = pick / 1000;
    }
    const double fl = 2.0 * d.M * d.N * d.K;
    flops_ += fl;
    d.workspace = ws_;
    d.workspace_bytes = ws_bytes_;
    d.fix_counters = fix_counters_;
    int tile = 0, splits = 1;
    (void)sdod_gemm_plan(&d, &tile, &splits);
    std::string label = "gemm_t" + std::to_string(tile); // one label per kernel symbol (tile table in gemm.hip); "xS" in the detail = split-K
    // algorithmic bytes: A read once (conv: the image, not its im2col), W once, out written once (+ residual read)
    double a_bytes = d.a_mode == SDOD_A_ROWS ? (double)d.M * d.K * 2
                                             : (double)d.n_img * d.h_in * d.w_in * (d.c0 + d.c1) * 2;
    double by = a_bytes + (double)d.N * d.K * 2 + (double)d.M * d.N * 2 * (d.residual ? 2 : 1);
    std::string detail = (d.a_mode == SDOD_A_ROWS ? std::string("rows") : "conv" + std::to_string(d.ksize) + (d.upsample ? "u" : "") +
                                                                          (d.stride == 2 ? "s2" : "") + (d.c1 ? "+cat" : "")) +
                         " M" + std::to_string(d.M) + " N" + std::to_string(d.N) + " K" + std::to_string(d.K) + " x" + std::to_string(splits);
    // In-kernel split-K reduce (gemm.hip: the last K slice to arrive at the tile's counter reduces; sdod_gemm_desc::fix_counters):
    // built, bit-identical to the reduce kernel, 31 launches fewer per evaluation -- and SLOWER: the evaluation's convolutions +
    // reduces take 1499 us against 1385 us (8-byte agent atomics: 1655 us): write-through partial stores, a fabric round trip
    // for the counter and a serial tail of (slices - 1) tile reads on one workgroup cost more than a reduce launch that runs on
    // the whole chip out of L2.  OFF unless SDOD_GEMM_FIXUP=1.
    static const bool fixup_on = [] { const char* e = std::getenv("SDOD_GEMM_FIXUP"); return e && e[0] == '1'; }();
    if (!fixup_on) d.fix_counters = nullptr;
    static const bool fuse_reduce_gn = [] { const char* e = std::getenv("SDOD_GN_REDUCE"); return e && e[0] == '1'; }();
    if (splits > 1 && !fuse_reduce_gn && sdod_gemm_fixup(&d)) {
        // the K slices are reduced inside the GEMM launch: ONE entry
        const double part = (double)splits * d.M * d.N * 4;
        sink().push_back(Op{[d](hipStream_t st) { check_rc(sdod_gemm_f16(&d, st)); }, label, fl, by + part, detail});
        return;
    }
    if (splits > 1) {
        // two launch-list entries, one per kernel, so that per-launch timings line up with rocprofv3's per-symbol numbers
        sdod_gemm_desc d1 = d, d2 = d;
        d1.phase = 1;
        d2.phase = 2;
        const double part = (double)splits * d.M * d.N * 4;
        sink().push_back(Op{[d1](hipStream_t st) { check_rc(sdod_gemm_f16(&d1, st)); }, label, fl, by + part, detail});
        if ((size_t)d.N * d.ldw * (d.wq ? 1 : 2) >= kPrefetchMinBytes) {
            sink().back().pf_ptr = d.w;
            sink().back().pf_bytes = (size_t)d.N * d.ldw * (d.wq ? 1 : 2);
        }
        // phase 2 is deferred: a GroupNorm that consumes d.out next folds it into its load (group_norm()), anything else
        // emits it as the stand-alone reduce launch
        pending_.active = true;
        pending_.d = d2;
        pending_.bytes = part + (double)d.M * d.N * 2 * (d.residual ? 2 : 1);
        pending_.detail = detail;
        // Measured on MI355X inside a replay (tools/gn_bench.py, profiles/r02_gn_bench.txt): the stand-alone reduce costs
        // 3.2-4.1 us and the one-launch GroupNorm 4.3-8.2 us, the fused form 7.6-16 us -- its 64 (image, group) workgroups
        // read the fp32 slabs with a quarter of the chip -- so fusing LOSES 0-3.7 us per site.  The deferral is therefore off
        // unless SDOD_GN_REDUCE=1 (kept, with its kernel and tests, for shapes where the balance differs).
        static const bool fuse_reduce = [] { const char* e = std::getenv("SDOD_GN_REDUCE"); return e && e[0] == '1'; }();
        if (to_static_ || !fuse_reduce) flush_pending();
        return;
    }
    sink().push_back(Op{[d](hipStream_t st) { check_rc(sdod_gemm_f16(&d, st)); }, label, fl, by, detail});
    if ((size_t)d.N * d.ldw * (d.wq ? 1 : 2) >= kPrefetchMinBytes) {
        sink().back().pf_ptr = d.w;
        sink().back().pf_bytes = (size_t)d.N * d.ldw * (d.wq ? 1 : 2);
    }
}

void Graph::linear_raw(const f16* x, int rows, int K, const f16* w, int ldw, int N, f16* out, const GemmOpt& o) {
    sdod_gemm_desc d{};
    d.a = x; d.w = w; d.out = out;
    d.M = rows; d.N = N; d.K = K;
    d.lda = o.lda ? o.lda : K;
    d.ldw = ldw;
    d.ldo = o.ldo ? o.ldo : N;
    d.a_mode = SDOD_A_ROWS;
    if (o.bias >= 0) d.bias = params_[o.bias].dev;
    if (o.bias_raw) d.bias = o.bias_raw;
    d.bias_on_m = o.bias_on_m ? 1 : 0;
    d.row_bias = o.row_bias; d.ld_row_bias = o.ld_row_bias; d.rows_per_img = o.rows_per_img;
    d.residual = o.residual; d.ldr = o.ldr ? o.ldr : d.ldo;
    d.act = o.act; d.alpha = o.alpha;
    if (o.geglu) {
        d.geglu = 1;
        if (!o.ldo) d.ldo = N / 2;
        d.ldr = d.ldo;
    }
    if (o.wq_scale) {
        d.wq = 1;
        d.w_scale = o.wq_scale;
        d.w_off = o.wq_off;
    }
    if (o.ln_w >= 0) {
        if (d.wq) throw Error(INTERNAL_ERROR, "LayerNorm fold with uint8 weights");
        // LayerNorm folded into this Linear: gamma goes into W (in place, once, after all parameters are set), the
        // kernel gathers the row statistics itself; s/t are derived vectors owned by the graph
        d.ln = 1;
        d.ln_eps = 1e-5f;
        if (mode_ == REAL) {
            const LnVecs v = ln_fold_vectors(const_cast<f16*>(w), N, K, ldw, o.ln_w, o.ln_b, reinterpret_cast<const float*>(d.bias));
            d.ln_s = v.s;
            d.bias = v.t;
        } else {
            d.ln_s = d.w; // placeholder pointers for the sizing pass
            d.bias = d.w;
        }
    }
    if (o.ln_s_raw) {
        d.ln = 1;
        d.ln_eps = 1e-5f;
        d.ln_s = o.ln_s_raw;
    }
    d.w_img_stride = o.w_img_stride; d.vec_img_stride = o.vec_img_stride; d.softmax_cols = o.softmax_cols;
    emit_gemm(d);
}

Graph::LnVecs Graph::ln_fold_vectors(f16* w, int N, int K, int ldw, int ln_w, int ln_b, const float* bias) {
    float *sv = nullptr, *tv = nullptr;
    SDOD_HIP_CHECK(hipMalloc((void**)&sv, (size_t)N * sizeof(float)));
    SDOD_HIP_CHECK(hipMalloc((void**)&tv, (size_t)N * sizeof(float)));
    derived_.push_back(sv);
    derived_.push_back(tv);
    fold_jobs_.push_back(FoldJob{w, N, K, ldw, W<float>(ln_w), W<float>(ln_b), bias, sv, tv});
    return LnVecs{sv, tv};
}

void Graph::linear(const f16* x, int rows, int K, int w, int N, f16* out, const GemmOpt& o) {
    const Param& p = params_[w];
    const int ldw = p.ld > 0 ? p.ld : p.kind == PK_CONV3_SMALL ? 64 : (int)(p.kind == PK_CONV3 ? p.shape[1] * 9 : p.shape[1]);
    if (p.quant) { // (ldw counts elements, i.e. bytes for the uint8 codes)
        GemmOpt oq = o;
        oq.wq_scale = qscale_of(w);
        oq.wq_off = qoff_of(w);
        linear_raw(x, rows, K, reinterpret_cast<const f16*>(p.dev), ldw, N, out, oq);
        return;
    }
    linear_raw(x, rows, K, reinterpret_cast<const f16*>(p.dev), ldw, N, out, o);
}

Act Graph::conv(const Act& x, const Act* x2, int w, int cout, int ksize, int stride, bool upsample, const GemmOpt& o) {
    const int ups = upsample ? 1 : 0;
    const int hup = x.h << ups, wup = x.w << ups;
    const int ho = (hup + 2 * (ksize / 2) - ksize) / stride + 1, wo = (wup + 2 * (ksize / 2) - ksize) / stride + 1;
    Act y;
    y.n = x.n; y.h = ho; y.w = wo; y.c = cout;
    y.p = o.out ? o.out : alloc(y.numel());
    sdod_gemm_desc d{};
    d.a = x.p; d.a2 = x2 ? x2->p : nullptr;
    d.w = params_[w].dev; d.out = y.p;
    const int cin = x.c + (x2 ? x2->c : 0);
    d.M = y.rows(); d.N = cout; d.K = ksize * ksize * cin;
    if (o.tail0) {
        d.k_tail = d.K;
        d.t0 = o.tail0->p; d.tc0 = o.tail0->c;
        d.t1 = o.tail1 ? o.tail1->p : nullptr; d.tc1 = o.tail1 ? o.tail1->c : 0;
        d.K += d.tc0 + d.tc1;
        if (o.bias2 >= 0) d.bias2 = params_[o.bias2].dev;
    }
    d.ldw = params_[w].ld > 0 ? params_[w].ld : d.K; d.ldo = cout;
    d.a_mode = SDOD_A_CONV3X3;
    d.n_img = x.n; d.h_in = x.h; d.w_in = x.w; d.c0 = x.c; d.c1 = x2 ? x2->c : 0;
    d.stride = stride; d.upsample = ups; d.ksize = ksize;
    if (o.bias >= 0) d.bias = params_[o.bias].dev;
    d.row_bias = o.row_bias; d.ld_row_bias = o.ld_row_bias; d.rows_per_img = o.rows_per_img;
    d.residual = o.residual; d.ldr = cout;
    d.act = o.act; d.alpha = o.alpha;
    if (params_[w].quant) {
        if (o.tail0) throw Error(INTERNAL_ERROR, "tail segment with uint8 weights");
        d.wq = 1;
        d.w_scale = qscale_of(w);
        d.w_off = qoff_of(w);
    }
    emit_gemm(d);
    return y;
}

Act Graph::group_norm(const Act& x, const Act* x2, int gw, int gb, float eps, bool silu) {
    Act y = act(x.n, x.h, x.w, x.c + (x2 ? x2->c : 0)); // allocated BEFORE parked tensors return to the arena (see settle())
    if (mode_ != DECLARE) gn_ws_need_ = std::max(gn_ws_need_, sdod_group_norm_workspace_bytes(x.n, 32));
    if (mode_ == REAL) {
        const void* xp = x.p; const void* x2p = x2 ? x2->p : nullptr; void* yp = y.p;
        const float* wp = W<float>(gw); const float* bp = W<float>(gb);
        const int n = x.n, hw = x.h * x.w, c0 = x.c, c1 = x2 ? x2->c : 0, si = silu ? 1 : 0;
        const std::string shape = "n" + std::to_string(n) + " hw" + std::to_string(hw) + " c" + std::to_string(c0 + c1);
        if (pending_.active && pending_.d.out == x.p && pending_.d.M == x.rows() && pending_.d.N == c0 &&
            (!pending_.d.row_bias || pending_.d.rows_per_img == hw) && sdod_group_norm_reduce_ok(hw, c0, c1, 32)) {
            // the producing conv is still in split-K form: this launch also performs its reduce + epilogue
            sdod_gn_reduce red{};
            check_rc(sdod_gemm_reduce_info(&pending_.d, &red));
            pending_.active = false;
            ops_.push_back(Op{[=](hipStream_t st) {
                check_rc(sdod_group_norm_reduce_nhwc(&red, x2p, yp, wp, bp, n, hw, c0, c1, 32, eps, si, st));
            }, "gn_group_red", 0, pending_.bytes + 2.0 * n * hw * (c0 + c1) * 2, shape + " x" + std::to_string(red.splits)});
        } else {
            flush_pending();
            void* ws = gn_ws_;
            ops_.push_back(Op{[=](hipStream_t st) {
                check_rc(sdod_group_norm_nhwc(xp, x2p, yp, wp, bp, n, hw, c0, c1, 32, eps, si, SDOD_F16, ws, st));
            }, [&] { // one label per kernel symbol, so that per-label timings line up with a profiler's per-symbol numbers
                switch (sdod_group_norm_path(n, hw, c0, c1, 32, SDOD_F16)) {
                case 0: return "gn_grid";
                case 1: return "gn_group";
                case 2: return "gn_small";
                default: return "gn_stats_apply";
                }
            }(), 0, 2.0 * n * hw * (c0 + c1) * 2, shape});
        }
    }
    drain_parked();
    return y;
}

Act Graph::layer_norm(const Act& x, int lw, int lb, float eps) {
    Act y = act(x.n, x.h, x.w, x.c);
    settle();
    if (mode_ != REAL) return y;
    const void* xp = x.p; void* yp = y.p;
    const float* wp = W<float>(lw); const float* bp = W<float>(lb);
    const int m = x.rows(), c = x.c;
    ops_.push_back(Op{[=](hipStream_t st) { check_rc(sdod_layer_norm_f16(xp, yp, wp, bp, m, c, eps, st)); }, "layer_norm", 0,
                      2.0 * m * c * 2, "m" + std::to_string(m) + " c" + std::to_string(c)});
    return y;
}

void Graph::attention(const f16* q, const f16* k, const f16* v, f16* out, int B, int heads, int lq, int lk, int d, int ldq,
                      int ldk, int ldv, int ldo, bool causal) {
    settle();
    if (mode_ != REAL) return;
    const double fl = 4.0 * B * heads * (double)lq * lk * d;
    flops_ += fl;
    const float scale = 1.0f / sqrtf((float)d);
    const int ca = causal ? 1 : 0;
    ops_.push_back(Op{[=](hipStream_t st) {
        check_rc(sdod_attention_f16(q, k, v, out, B, heads, lq, lk, d, ldq, ldk, ldv, ldo, scale, ca, st));
    }, "attn_d" + std::to_string(d), fl, 2.0 * B * heads * d * (2.0 * lq + 2.0 * lk),
                      "B" + std::to_string(B) + " h" + std::to_string(heads) + " lq" + std::to_string(lq) + " lk" + std::to_string(lk)});
}

// ------------------------------------------------------------------------------------------ finalize / run
void Graph::build() {
    quant_global(); // (the UNet builder narrows it block by block: quant_rows())
    switch (kind_) {
    case SDOD_GRAPH_UNET: build_unet(); break;
    case SDOD_GRAPH_VAE_DECODER: build_vae(); break;
    case SDOD_GRAPH_TEXT_ENCODER: build_clip(); break;
    default: build_temb(); break;
    }
    settle();
}

void Graph::finalize() {
    allocate_weights();
    SDOD_REQUIRE(!finalized_, "graph already finalized");
    std::string missing;
    int nmiss = 0;
    for (auto& p : params_)
        if (!p.set) {
            if (nmiss < 5) missing += (nmiss ? ", " : "") + p.name;
            ++nmiss;
        }
    SDOD_REQUIRE(nmiss == 0, std::to_string(nmiss) + " parameter(s) not set: " + missing + (nmiss > 5 ? ", ..." : ""));
    mode_ = DRY;
    arena_reset();
    ws_need_ = gn_ws_need_ = 0;
    build();
    arena_cap_ = align_up(arena_high_, 256) + 256;
    ws_bytes_ = align_up(ws_need_, 256) + 256;
    gn_ws_bytes_ = align_up(gn_ws_need_, 256) + 256;
    SDOD_HIP_CHECK(hipMalloc((void**)&arena_base_, arena_cap_));
    SDOD_HIP_CHECK(hipMalloc((void**)&ws_, ws_bytes_));
    SDOD_HIP_CHECK(hipMalloc((void**)&gn_ws_, gn_ws_bytes_));
    SDOD_HIP_CHECK(hipMemset(gn_ws_, 0, gn_ws_bytes_)); // the one-launch GroupNorm keeps its grid-barrier words in here: zero once
    SDOD_HIP_CHECK(hipMalloc((void**)&fix_counters_, sdod_gemm_fixup_counters() * sizeof(unsigned)));
    SDOD_HIP_CHECK(hipMemset(fix_counters_, 0, sdod_gemm_fixup_counters() * sizeof(unsigned)));
    if (kind_ == SDOD_GRAPH_UNET && kv_total_ > 0)
        SDOD_HIP_CHECK(hipMalloc((void**)&kv_all_, (size_t)batch_ * cfg_.context_len * kv_total_ * sizeof(f16)));
    mode_ = REAL;
    arena_reset();
    ops_.clear();
    static_ops_.clear();
    fold_jobs_.clear();
    compose_jobs_.clear();
    flops_ = 0;
    build();
    for (const FoldJob& j : fold_jobs_)
        check_rc(sdod_ln_fold_f16(j.w, j.n, j.k, j.ldw, j.gamma, j.beta, j.bias_in, j.s, j.t, nullptr));
    for (const ComposeJob& j : compose_jobs_) {
        // P . W needs every row of W for every output row: compose from a copy of the W block
        f16* tmp = nullptr;
        SDOD_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)j.n_mid * j.k * sizeof(f16)));
        SDOD_HIP_CHECK(hipMemcpy2D(tmp, (size_t)j.k * sizeof(f16), j.c, (size_t)j.ld * sizeof(f16), (size_t)j.k * sizeof(f16), (size_t)j.n_mid,
                                   hipMemcpyDeviceToDevice));
        const int rc = sdod_compose_linear_f16(j.p, j.ld, tmp, j.k, j.c, j.ld, j.n_out, j.n_mid, j.k, j.bias_w, j.bias_p, j.bias_out, nullptr);
        SDOD_HIP_CHECK(hipDeviceSynchronize());
        (void)hipFree(tmp);
        check_rc(rc);
    }
    SDOD_HIP_CHECK(hipDeviceSynchronize());
    if (tune_scratch()) {
        (void)hipFree(tune_scratch());
        tune_scratch() = nullptr;
    }
    finalized_ = true;
}

// ---- weight prefetch.  Inside a replay every weight matrix comes from HBM (1.7 GB of weights sweep the 256 MiB Infinity Cache
// many times per evaluation) and the deep, weight-heavy GEMMs -- 30-60 MB of weights for a few microseconds of arithmetic on
// a 16x16 or 8x8 map -- are bound by the bytes they can keep in flight: timed with their weights cache-resident they run
// 25-35 % faster (profiles/r02_deep_conv_hot_cold.txt).  HBM itself is ~95 % idle over the evaluation, so a tiny kernel on a
// side stream touches every 128-byte line of such a matrix a few launches ahead of its consumer (a parallel branch of the
// captured graph: fork by event, one join behind the last launch).
// MEASURED: the branch costs far more than it brings -- 8.04 vs 9.56 images/s on one box (bench.py, SDOD_PREFETCH=1 vs unset):
// like the decode / sampling overlap, anything running next to the latency-bound main chain slows it down.  OFF unless
// SDOD_PREFETCH=1; kept as a documented experiment.
namespace {
constexpr int kPrefetchLookahead = 3;
bool prefetch_enabled() {
    static const bool on = [] { const char* e = std::getenv("SDOD_PREFETCH"); return e && e[0] == '1'; }();
    return on;
}
} // namespace

void Graph::run_ops(hipStream_t st) {
    const bool pf = prefetch_enabled() && g_launch_timer == nullptr;
    size_t ev = 0;
    bool forked = false;
    auto next_event = [&]() {
        if (ev == pf_events_.size()) {
            hipEvent_t e = nullptr;
            SDOD_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            pf_events_.push_back(e);
        }
        return pf_events_[ev++];
    };
    for (size_t i = 0; i < ops_.size(); ++i) {
        const size_t j = i + kPrefetchLookahead;
        if (pf && j < ops_.size() && ops_[j].pf_bytes) {
            if (!side_stream_) SDOD_HIP_CHECK(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
            hipEvent_t fork = next_event();
            SDOD_HIP_CHECK(hipEventRecord(fork, st));
            SDOD_HIP_CHECK(hipStreamWaitEvent(side_stream_, fork, 0));
            check_rc(sdod_l2_prefetch(ops_[j].pf_ptr, ops_[j].pf_bytes, side_stream_));
            forked = true;
        }
        ops_[i].fn(st);
    }
    if (forked) { // the branch rejoins behind the last launch (a capture must end with every forked stream joined)
        hipEvent_t join = next_event();
        SDOD_HIP_CHECK(hipEventRecord(join, side_stream_));
        SDOD_HIP_CHECK(hipStreamWaitEvent(st, join, 0));
    }
}

void Graph::check_health() const {
    // the one-launch GroupNorm's grid barrier gives up after ~1 s rather than hang the device; what it then wrote is garbage
    // and it says so through a sticky host-visible word (norms.hip).  Read here: at the next execute and wherever a caller
    // has just synchronised (sdod_graph_check, Context::generate behind the image copy).
    if (sdod_group_norm_status() != 0)
        throw Error(RUNTIME_ERROR, "a GroupNorm grid barrier timed out on this device (workspace clobbered, or graphs sharing the device "
                                   "starved each other of CUs): results since then are invalid; sdod_group_norm_clear_error() after "
                                   "rebuilding the graph re-arms");
}

void Graph::execute(hipStream_t st, bool use_hip_graph, bool skip_static) {
    SDOD_REQUIRE(finalized_, "graph not finalized");
    check_health();
    if (!skip_static || eager_runs_ == 0)
        for (auto& op : static_ops_) op.fn(st);
    if (!use_hip_graph || eager_runs_ == 0) {
        // the first run is always eager: it sets kernel attributes (dynamic LDS sizes), which must not happen in capture
        run_ops(st);
        ++eager_runs_;
        return;
    }
    if (!graph_exec_) {
        // capture on a private stream (the caller's may be the legacy default stream, which cannot capture);
        // kernel nodes carry no stream identity, so the instantiated graph replays on any stream
        if (!capture_stream_) SDOD_HIP_CHECK(hipStreamCreateWithFlags(&capture_stream_, hipStreamNonBlocking));
        SDOD_HIP_CHECK(hipStreamSynchronize(st));
        SDOD_HIP_CHECK(hipStreamBeginCapture(capture_stream_, hipStreamCaptureModeThreadLocal));
        try {
            run_ops(capture_stream_);
        } catch (...) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(capture_stream_, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        SDOD_HIP_CHECK(hipStreamEndCapture(capture_stream_, &hip_graph_));
        SDOD_HIP_CHECK(hipGraphInstantiate(&graph_exec_, hip_graph_, nullptr, nullptr, 0));
    }
    SDOD_HIP_CHECK(hipGraphLaunch(graph_exec_, st));
}

void Graph::profile(hipStream_t st, int iters, float* ms, int n) {
    SDOD_REQUIRE(finalized_, "graph not finalized");
    SDOD_REQUIRE(ms != nullptr && n == (int)ops_.size() && iters > 0, "profile buffer must hold one float per op");
    // every launch-list entry is run eagerly with a LaunchTimer installed: its kernels (one; two or three for the two-launch
    // GroupNorm) are stamped at their own begin / end by the command processor, i.e. the per-dispatch duration a profiler
    // reports -- no queue latency inside the figure
    constexpr int kMaxLaunches = 4;
    struct Events { // destroyed on every exit path, an op that throws included
        std::vector<hipEvent_t> v;
        explicit Events(size_t n) : v(n, nullptr) {}
        ~Events() { for (auto e : v) if (e) (void)hipEventDestroy(e); }
        hipEvent_t& operator[](size_t i) { return v[i]; }
    } evs(ops_.size() * kMaxLaunches), eve(ops_.size() * kMaxLaunches);
    for (auto& e : evs.v) SDOD_HIP_CHECK(hipEventCreate(&e));
    for (auto& e : eve.v) SDOD_HIP_CHECK(hipEventCreate(&e));
    std::vector<int> used(ops_.size(), 0);
    std::vector<std::vector<float>> samples(ops_.size());
    for (int it = 0; it < iters + 1; ++it) { // first pass is a warm-up
        for (size_t i = 0; i < ops_.size(); ++i) {
            LaunchTimer lt{&evs[i * kMaxLaunches], &eve[i * kMaxLaunches], kMaxLaunches, 0};
            g_launch_timer = &lt;
            try {
                ops_[i].fn(st);
            } catch (...) {
                g_launch_timer = nullptr;
                throw;
            }
            g_launch_timer = nullptr;
            used[i] = lt.used;
        }
        SDOD_HIP_CHECK(hipStreamSynchronize(st));
        if (it == 0) continue;
        for (size_t i = 0; i < ops_.size(); ++i) {
            float tot = 0.f;
            for (int k = 0; k < used[i]; ++k) {
                float t = 0.f;
                SDOD_HIP_CHECK(hipEventElapsedTime(&t, evs[i * kMaxLaunches + k], eve[i * kMaxLaunches + k]));
                tot += t;
            }
            samples[i].push_back(tot);
        }
    }
    // median over the passes: the start stamp is a marker in front of the kernel, so a host hiccup between the two enqueues
    // lands inside one sample; it must not move the figure
    for (size_t i = 0; i < ops_.size(); ++i) {
        std::sort(samples[i].begin(), samples[i].end());
        const size_t k = samples[i].size();
        ms[i] = k == 0 ? 0.f : (k & 1) ? samples[i][k / 2] : 0.5f * (samples[i][k / 2 - 1] + samples[i][k / 2]);
    }
    ++eager_runs_;
}

void Graph::stats(size_t* wbytes, size_t* abytes, int* launches, double* flops) const {
    if (wbytes) *wbytes = weight_bytes_;
    if (abytes) *abytes = arena_cap_;
    if (launches) *launches = (int)ops_.size();  // per evaluation; static_ops_ run once per static-input change
    if (flops) *flops = flops_;
}

} // namespace sdod

// ================================================================================================ C ABI
using sdod::Graph;

extern "C" void sdod_model_config_sd14(sdod_model_config* cfg) {
    if (!cfg) return;
    cfg->latent_channels = 4;
    cfg->latent_h = 64;
    cfg->latent_w = 64;
    cfg->model_channels = 320;
    cfg->context_dim = 768;
    cfg->context_len = 77;
    cfg->num_heads = 8;
    cfg->head_dim = 0;
    cfg->vocab_size = 49408;
    cfg->text_layers = 12;
    cfg->text_heads = 12;
    cfg->vae_channels = 128;
    cfg->linear_proj = 0;
    cfg->text_arch = 0;
}

extern "C" void sdod_model_config_sd21(sdod_model_config* cfg) {
    if (!cfg) return;
    sdod_model_config_sd14(cfg);
    cfg->latent_h = cfg->latent_w = 96;
    cfg->context_dim = 1024;
    cfg->num_heads = 0;
    cfg->head_dim = 64;
    cfg->linear_proj = 1;
    cfg->text_arch = 1;
    cfg->text_layers = 23;
    cfg->text_heads = 16;
}

extern "C" int sdod_graph_create(void** graph, int kind, const sdod_model_config* cfg, int batch) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr && cfg != nullptr, "null argument");
    *graph = nullptr;
    *graph = new Graph(kind, *cfg, batch);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_destroy(void* graph) {
    SDOD_TRY
    delete static_cast<Graph*>(graph);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_num_params(void* graph) { return graph ? static_cast<Graph*>(graph)->num_params() : 0; }

extern "C" int sdod_graph_param_info(void* graph, int index, const char** name, int* ndim, int64_t shape[4]) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    auto* g = static_cast<Graph*>(graph);
    SDOD_REQUIRE(index >= 0 && index < g->num_params(), "parameter index out of range");
    const sdod::Param& p = g->param(index);
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int)p.shape.size();
    if (shape)
        for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_set_param(void* graph, const char* name, const void* data, int dtype, const int64_t* shape, int ndim) {
    SDOD_TRY
    SDOD_REQUIRE(graph && name && shape, "null argument");
    static_cast<Graph*>(graph)->set_param(name, data, dtype, shape, ndim);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_load_file(void* graph, const char* path, const char* prefix) {
    SDOD_TRY
    SDOD_REQUIRE(graph && path, "null argument");
    static_cast<Graph*>(graph)->load_file(path, prefix ? prefix : "");
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_finalize(void* graph) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    static_cast<Graph*>(graph)->finalize();
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_io(void* graph, int is_output, int index, void** device_ptr, size_t* bytes) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    const sdod::IoSlot s = static_cast<Graph*>(graph)->io(is_output != 0, index);
    if (device_ptr) *device_ptr = s.ptr;
    if (bytes) *bytes = s.bytes;
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_execute(void* graph, void* stream, int use_hip_graph) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    static_cast<Graph*>(graph)->execute((hipStream_t)stream, (use_hip_graph & 1) != 0, (use_hip_graph & 2) != 0);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_check(void* graph) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    static_cast<Graph*>(graph)->check_health();
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_num_ops(void* graph) { return graph ? static_cast<Graph*>(graph)->num_ops() : 0; }

extern "C" int sdod_graph_op_detail(void* graph, int index, const char** detail) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr && detail != nullptr, "null argument");
    auto* g = static_cast<Graph*>(graph);
    SDOD_REQUIRE(index >= 0 && index < g->num_ops(), "op index out of range");
    *detail = g->op(index).detail.c_str();
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_op_info(void* graph, int index, const char** label, double* flops, double* bytes) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    auto* g = static_cast<Graph*>(graph);
    SDOD_REQUIRE(index >= 0 && index < g->num_ops(), "op index out of range");
    const auto& op = g->op(index);
    if (label) *label = op.label.c_str();
    if (flops) *flops = op.flops;
    if (bytes) *bytes = op.bytes;
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_profile(void* graph, void* stream, int iters, float* ms_out, int n) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    static_cast<Graph*>(graph)->profile((hipStream_t)stream, iters, ms_out, n);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_tune_info(void* graph, int* from_table, int* tuned_in_process, char* table_path, int cap) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    auto* g = static_cast<Graph*>(graph);
    if (from_table) *from_table = g->tune_hits();
    if (tuned_in_process) *tuned_in_process = g->tune_misses();
    if (table_path && cap > 0) {
        std::string s = sdod::shipped_tune_path();
        if (const char* extra = sdod::tune_cache_path()) s += (s.empty() ? "" : " + ") + std::string(extra);
        std::snprintf(table_path, (size_t)cap, "%s", s.c_str());
    }
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_graph_stats(void* graph, size_t* weight_bytes, size_t* arena_bytes, int* num_launches, double* flops) {
    SDOD_TRY
    SDOD_REQUIRE(graph != nullptr, "null graph");
    static_cast<Graph*>(graph)->stats(weight_bytes, arena_bytes, num_launches, flops);
    return 0;
    SDOD_CATCH
}
