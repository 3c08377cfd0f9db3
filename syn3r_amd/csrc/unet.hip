// Single-call UNet forward at the C-ABI: syn3r_unet_create / _workspace_bytes / _forward / _destroy.
//
// What SURVEY.md 8(b) lists for hosts that are not Python: `self.unet(latent_model_input, t, encoder_hidden_states=...,
// added_time_ids=..., return_dict=False)[0]` (model/SVD_2pass_prob_uncertain_post.py:763,786;
// diffusers/models/unets/unet_spatio_temporal_condition.py:356-489) as ONE call on device pointers, with the weights read
// from a diffusers checkpoint directory (`from_pretrained(<local dir>, torch_dtype=float16, variant="fp16")`,
// model/diffusionGS.py:1089).  This file is the host graph only - the same launch sequence, on the same operators of this
// library (gemm.hip, attn.hip, norm.hip), that syn3r_amd/unet/model.py issues from Python: safetensors reader, weight
// repacking (OHWI convolutions, fused qkv, GEGLU row groups, stacked time-embedding projections), a stream-ordered arena
// over the caller's workspace, and the block loops of unet_3d_blocks.py / transformer_temporal.py / resnet.py.
// The handful of elementwise steps the Python host leaves to torch (sinusoidal embeddings, SiLU on [B, 1280] vectors,
// NCHW <-> NHWC at the two ends) are small kernels at the top of the file.
#include "common.h"
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <math.h>
#include <string.h>
#include <ctype.h>
#include <exception>
#include <map>
#include <memory>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <mutex>
#include <vector>

using namespace syn3r;

namespace {

typedef unsigned short u16;

// ------------------------------------------------------------------------------------------------ glue kernels
// embeddings.py Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of t * exp(-ln(1e4) i / half),
// fp32 arithmetic in the order model.py:timestep_embedding spells it, rounded to fp16 at the end.
__global__ void k_timestep_embedding(const float* __restrict__ t_dev, double t_host, float t_step, int n, int dim, __half* __restrict__ out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int r = i / half, c = i - r * half;
    const float t = t_dev ? t_dev[r] : (float)t_host + t_step * (float)r;
    // (`tensor / python_int` on the device is a multiplication by the fp32 reciprocal in torch: the same here, or the two hosts
    // differ in the last bit of the exponent whenever `half` is not a power of two)
    const float e = expf((-9.210340371976184f * (float)c) * (1.0f / (float)half));
    const float a = t * e;
    out[(size_t)r * dim + c] = __float2half(cosf(a));
    out[(size_t)r * dim + half + c] = __float2half(sinf(a));
}
__global__ void k_silu(const __half* __restrict__ x, __half* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = __half2float(x[i]);
    y[i] = __float2half(v / (1.0f + expf(-v)));
}
__global__ void k_add(const __half* __restrict__ a, const __half* __restrict__ b, __half* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __float2half(__half2float(a[i]) + __half2float(b[i]));
}
// [N, C, h, w] -> [N, h, w, CP] (channels zero-padded to CP)
__global__ void k_nchw_to_nhwc(const __half* __restrict__ x, __half* __restrict__ y, int N, int C, int hw, int CP) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * hw * CP) return;
    const int c = (int)(i % CP);
    const long long pix = i / CP;
    const int n = (int)(pix / hw), p = (int)(pix - (long long)n * hw);
    y[i] = c < C ? x[((long long)n * C + c) * hw + p] : __float2half(0.f);
}
// [N, h, w, CP] -> [N, C, h, w] (the first C channels)
__global__ void k_nhwc_to_nchw(const __half* __restrict__ x, __half* __restrict__ y, int N, int C, int hw, int CP) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * C * hw) return;
    const int p = (int)(i % hw);
    const long long nc = i / hw;
    const int n = (int)(nc / C), c = (int)(nc - (long long)n * C);
    y[i] = x[((long long)n * hw + p) * CP + c];
}
// y[r] = x[r % rows_in]  (e.repeat(B, 1))
__global__ void k_repeat_rows(const __half* __restrict__ x, __half* __restrict__ y, int rows_in, long long rows_out, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_out * C) return;
    const long long r = i / C;
    y[i] = x[(r % rows_in) * C + (i - r * C)];
}
// [x1 | x2] along the channels (the fallback of the two-source contraction)
__global__ void k_cat_cols(const __half* __restrict__ a, int C1, const __half* __restrict__ b, int C2, __half* __restrict__ y, long long M) {
    const int C = C1 + C2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    y[i] = c < C1 ? a[r * C1 + c] : b[r * C2 + (c - C1)];
}

// ------------------------------------------------------------------------------------------------ JSON (config.json, safetensors header)
struct JVal {
    enum Type { NUL, NUM, STR, ARR, OBJ, BOOL } type = NUL;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const {
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
struct JParser {
    const char* p; const char* e; bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool lit(const char* s) { size_t n = strlen(s); if ((size_t)(e - p) >= n && !memcmp(p, s, n)) { p += n; return true; } return false; }
    std::string string_() {
        std::string s;
        if (p >= e || *p != '"') { ok = false; return s; }
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'u':                                  // \uXXXX: not needed for tensor names; the four digits must exist
                        if (e - p < 5) { ok = false; return s; }
                        s += '?'; p += 4; break;
                    default: s += *p;
                }
                ++p;
            } else s += *p++;
        }
        if (p >= e) { ok = false; return s; }
        ++p;
        return s;
    }
    JVal value(int depth = 0) {
        JVal v;
        ws();
        if (p >= e || depth > 64) { ok = false; return v; }
        if (*p == '{') {
            v.type = JVal::OBJ; ++p; ws();
            if (p < e && *p == '}') { ++p; return v; }
            while (ok) {
                ws();
                std::string k = string_();
                ws();
                if (p >= e || *p != ':') { ok = false; break; }
                ++p;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; break; }
                ok = false;
            }
        } else if (*p == '[') {
            v.type = JVal::ARR; ++p; ws();
            if (p < e && *p == ']') { ++p; return v; }
            while (ok) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; break; }
                ok = false;
            }
        } else if (*p == '"') { v.type = JVal::STR; v.str = string_(); }
        else if (lit("true")) { v.type = JVal::BOOL; v.num = 1; }
        else if (lit("false")) { v.type = JVal::BOOL; v.num = 0; }
        else if (lit("null")) { v.type = JVal::NUL; }
        else {
            // the header is a slice of an mmap'd file: not NUL-terminated, so the number is parsed from a bounded copy
            char buf[64];
            size_t n = 0;
            while (p + n < e && n < sizeof(buf) - 1 && (isdigit((unsigned char)p[n]) || p[n] == '-' || p[n] == '+' || p[n] == '.' || p[n] == 'e' || p[n] == 'E')) ++n;
            memcpy(buf, p, n);
            buf[n] = 0;
            char* end = nullptr;
            v.type = JVal::NUM; v.num = n ? strtod(buf, &end) : 0.0;
            if (n == 0 || end != buf + n || !(v.num == v.num)) ok = false; else p += n;
        }
        return v;
    }
};

bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

// ------------------------------------------------------------------------------------------------ safetensors
struct StEntry { std::string dtype; std::vector<long long> shape; size_t begin = 0, end = 0; };
struct StFile {
    int fd = -1; const unsigned char* map = nullptr; size_t size = 0, data0 = 0;
    std::unordered_map<std::string, StEntry> entries;
    ~StFile() { if (map) munmap((void*)map, size); if (fd >= 0) close(fd); }
    bool open_(const std::string& path, std::string& err) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "cannot open " + path; return false; }
        struct stat sb;
        if (fstat(fd, &sb) || sb.st_size < 8) { err = path + ": not a safetensors file"; return false; }
        size = (size_t)sb.st_size;
        map = (const unsigned char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map == MAP_FAILED) { map = nullptr; err = "mmap failed on " + path; return false; }
        unsigned long long hl = 0;
        memcpy(&hl, map, 8);
        if (hl == 0 || hl > size - 8) { err = path + ": bad header length"; return false; }
        data0 = 8 + (size_t)hl;
        JParser jp{(const char*)map + 8, (const char*)map + 8 + hl};
        JVal h = jp.value();
        if (!jp.ok || h.type != JVal::OBJ) { err = path + ": header is not JSON"; return false; }
        for (auto& kv : h.obj) {
            if (kv.first == "__metadata__") continue;
            const JVal *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
            if (!dt || !sh || !off || off->arr.size() != 2) { err = path + ": malformed entry " + kv.first; return false; }
            StEntry en;
            en.dtype = dt->str;
            // dimensions and offsets are file contents: integers in [0, 2^53) only, the element count bounded by the file size
            auto whole = [](const JVal& x) { return x.type == JVal::NUM && x.num >= 0 && x.num < 9007199254740992.0 && x.num == (double)(unsigned long long)x.num; };
            if (sh->type != JVal::ARR || sh->arr.size() > 8 || !whole(off->arr[0]) || !whole(off->arr[1])) { err = path + ": malformed entry " + kv.first; return false; }
            unsigned long long count = 1;
            for (auto& d : sh->arr) {
                if (!whole(d)) { err = path + ": bad shape in entry " + kv.first; return false; }
                if (__builtin_mul_overflow(count, (unsigned long long)d.num, &count) || count > (unsigned long long)size) { err = path + ": shape of entry " + kv.first + " exceeds the file"; return false; }
                en.shape.push_back((long long)d.num);
            }
            en.begin = (size_t)off->arr[0].num; en.end = (size_t)off->arr[1].num;
            if (en.end < en.begin || en.end > size - data0) { err = path + ": entry " + kv.first + " outside the file"; return false; }
            entries.emplace(kv.first, std::move(en));
        }
        return true;
    }
};

inline u16 f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; u16 b; memcpy(&b, &h, 2); return b; }
inline float f16_bits_to_f32(u16 b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }

// a checkpoint tensor as fp16 bits on the host (F16 as stored; F32 / BF16 rounded once, as `.to(torch.float16)` does)
bool host_f16(const StFile& f, const StEntry& e, std::vector<u16>& out, std::string& err) {
    unsigned long long cnt = 1;
    for (long long d : e.shape)
        if (d < 0 || __builtin_mul_overflow(cnt, (unsigned long long)d, &cnt)) { err = "bad shape"; return false; }
    const size_t esz = e.dtype == "F32" ? 4 : ((e.dtype == "F16" || e.dtype == "BF16") ? 2 : 0);
    if (!esz) { err = "unsupported dtype " + e.dtype; return false; }
    if (cnt > (e.end - e.begin) / esz || cnt * esz != e.end - e.begin) { err = "size mismatch"; return false; }   // before any allocation
    const long long n = (long long)cnt;
    const unsigned char* src = f.map + f.data0 + e.begin;
    out.resize((size_t)n);
    if (e.dtype == "F16") {
        memcpy(out.data(), src, (size_t)n * 2);
    } else if (e.dtype == "F32") {
        if ((size_t)n * 4 != e.end - e.begin) { err = "size mismatch"; return false; }
        for (long long i = 0; i < n; ++i) { float v; memcpy(&v, src + 4 * i, 4); out[(size_t)i] = f32_to_f16_bits(v); }
    } else if (e.dtype == "BF16") {
        if ((size_t)n * 2 != e.end - e.begin) { err = "size mismatch"; return false; }
        for (long long i = 0; i < n; ++i) { unsigned int u = 0; u16 b; memcpy(&b, src + 2 * i, 2); u = (unsigned)b << 16; float v; memcpy(&v, &u, 4); out[(size_t)i] = f32_to_f16_bits(v); }
    } else { err = "unsupported dtype " + e.dtype; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------ model
struct Wt { __half* p = nullptr; long long rows = 0, cols = 0; };       // a packed device tensor ([rows, cols], cols = the rest)
struct BlockPlan { int idx; bool attn; std::vector<std::pair<int, int>> layers; bool resample; int ch, heads; };

}  // namespace

struct syn3r_unet {
    int device = 0;
    int in_ch = 8, out_ch = 4, add_dim = 256, proj_in = 768;
    std::vector<int> boc, heads, layers, cross;
    std::vector<BlockPlan> down, up;
    char* blob = nullptr; size_t blob_bytes = 0;                        // every packed weight, one allocation
    std::unordered_map<std::string, Wt> w;
    std::unordered_map<std::string, std::pair<double, double>> alpha;   // mix_factor -> (alpha, 1 - alpha) as the fp16 values the reference uses
    std::unordered_map<std::string, std::pair<int, int>> temb_slice;
    std::vector<std::string> temb_names;
    struct PosEmb { __half* p = nullptr; hipEvent_t ready = nullptr; hipStream_t stream = nullptr; };
    std::map<std::string, PosEmb> pos_cache;                            // frame-position embeddings per (block, F, B): functions of the weights only
    bool ff_ln = true, ln_qkv = true;
};

namespace {

// Live handles: every entry point checks its handle against this set (a stale or fabricated pointer is an error, not a crash).
std::mutex g_unet_mu;
std::unordered_set<const syn3r_unet*>& unet_registry() { static auto* s = new std::unordered_set<const syn3r_unet*>; return *s; }
bool unet_live(const syn3r_unet* m) {
    std::lock_guard<std::mutex> lk(g_unet_mu);
    return m && unet_registry().count(m);
}

struct HostPack { std::string name; long long rows, cols; std::vector<u16> d; };

int fail(const char* fmt, const std::string& a) { set_error(fmt, a.c_str()); return SYN3R_E_INVALID; }

std::vector<int> int_list(const JVal* v, size_t n, int dflt) {
    std::vector<int> r;
    if (!v) return std::vector<int>(n, dflt);
    if (v->type == JVal::ARR) { for (auto& x : v->arr) r.push_back((int)x.num); }
    else r.assign(n, (int)v->num);
    return r;
}

// model.py:_declare - the block plan of unet_spatio_temporal_condition.py:140-330
int build_plan(syn3r_unet& m, const JVal& cfg) {
    auto num = [&](const char* k, int d) { const JVal* v = cfg.get(k); return v && v->type == JVal::NUM ? (int)v->num : d; };
    m.in_ch = num("in_channels", 8); m.out_ch = num("out_channels", 4);
    m.add_dim = num("addition_time_embed_dim", 256); m.proj_in = num("projection_class_embeddings_input_dim", 768);
    const JVal* b = cfg.get("block_out_channels");
    m.boc = b ? int_list(b, 0, 0) : std::vector<int>{320, 640, 1280, 1280};
    const size_t n = m.boc.size();
    SYN3R_REQUIRE(n >= 1 && n <= 8, "unet_create: %zu levels", n);
    m.heads = cfg.get("num_attention_heads") ? int_list(cfg.get("num_attention_heads"), n, 0) : std::vector<int>{5, 10, 20, 20};
    m.layers = int_list(cfg.get("layers_per_block"), n, 2);
    m.cross = int_list(cfg.get("cross_attention_dim"), n, 1024);
    std::vector<int> tl = int_list(cfg.get("transformer_layers_per_block"), n, 1);
    SYN3R_REQUIRE(m.heads.size() == n && m.layers.size() == n && m.cross.size() == n && tl.size() == n, "unet_create: per-level lists of config.json do not match block_out_channels");
    for (size_t i = 0; i < n; ++i) {
        SYN3R_REQUIRE(tl[i] == 1, "unet_create: transformer_layers_per_block != 1 is not used by SVD");
        SYN3R_REQUIRE(m.boc[i] % 64 == 0 && m.boc[i] == 64 * m.heads[i], "unet_create: attention head dim must be 64 (level %zu: %d channels, %d heads)", i, m.boc[i], m.heads[i]);
        SYN3R_REQUIRE(m.layers[i] >= 1 && m.layers[i] <= 8, "unet_create: layers_per_block");
    }
    std::vector<std::string> dt, ut;
    if (const JVal* v = cfg.get("down_block_types")) for (auto& x : v->arr) dt.push_back(x.str);
    else { dt.assign(n, "CrossAttnDownBlockSpatioTemporal"); dt.back() = "DownBlockSpatioTemporal"; }
    if (const JVal* v = cfg.get("up_block_types")) for (auto& x : v->arr) ut.push_back(x.str);
    else { ut.assign(n, "CrossAttnUpBlockSpatioTemporal"); ut.front() = "UpBlockSpatioTemporal"; }
    SYN3R_REQUIRE(dt.size() == n && ut.size() == n, "unet_create: block type lists do not match block_out_channels");
    int out_ch = m.boc[0];
    for (size_t i = 0; i < n; ++i) {
        const int in_ch = out_ch;
        out_ch = m.boc[i];
        const bool attn = dt[i] == "CrossAttnDownBlockSpatioTemporal";
        if (!attn && dt[i] != "DownBlockSpatioTemporal") return fail("unet_create: %s does not exist", dt[i]);
        BlockPlan bp{(int)i, attn, {}, i != n - 1, out_ch, m.heads[i]};
        for (int j = 0; j < m.layers[i]; ++j) bp.layers.emplace_back(j == 0 ? in_ch : out_ch, out_ch);
        m.down.push_back(bp);
    }
    std::vector<int> rev(m.boc.rbegin(), m.boc.rend()), rev_heads(m.heads.rbegin(), m.heads.rend()), rev_layers(m.layers.rbegin(), m.layers.rend());
    out_ch = rev[0];
    for (size_t i = 0; i < n; ++i) {
        const bool attn = ut[i] == "CrossAttnUpBlockSpatioTemporal";
        if (!attn && ut[i] != "UpBlockSpatioTemporal") return fail("unet_create: %s does not exist", ut[i]);
        const int prev_out = out_ch;
        out_ch = rev[i];
        const int in_ch = rev[std::min(i + 1, n - 1)];
        const int nl = rev_layers[i] + 1;
        BlockPlan bp{(int)i, attn, {}, i != n - 1, out_ch, rev_heads[i]};
        for (int j = 0; j < nl; ++j) {
            const int res_skip = j == nl - 1 ? in_ch : out_ch, res_in = j == 0 ? prev_out : out_ch;
            bp.layers.emplace_back(res_in + res_skip, out_ch);
        }
        m.up.push_back(bp);
    }
    return SYN3R_OK;
}

// model.py:_pack - the kernel-side layouts, built on the host from the checkpoint's tensors
int pack_weights(syn3r_unet& m, const StFile& f, std::vector<HostPack>& packs) {
    std::string err;
    auto has_suffix = [](const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && !s.compare(s.size() - n, n, suf); };
    std::map<std::string, const StEntry*> names;                         // sorted: a deterministic blob layout
    for (auto& kv : f.entries) names[kv.first] = &kv.second;
    auto get = [&](const std::string& k, std::vector<u16>& out, std::vector<long long>* shape = nullptr) -> bool {
        auto it = f.entries.find(k);
        if (it == f.entries.end()) { err = "missing tensor " + k; return false; }
        if (shape) *shape = it->second.shape;
        return host_f16(f, it->second, out, err);
    };
    // load_state_dict's order is the DECLARATION order; the stacked time-embedding projection follows it.  Recreate that order.
    auto res_names = [&](const std::string& pre, std::vector<std::string>& out) {
        out.push_back(pre + ".spatial_res_block.time_emb_proj");
        out.push_back(pre + ".temporal_res_block.time_emb_proj");
    };
    for (auto& bp : m.down) for (size_t j = 0; j < bp.layers.size(); ++j) res_names("down_blocks." + std::to_string(bp.idx) + ".resnets." + std::to_string(j), m.temb_names);
    res_names("mid_block.resnets.0", m.temb_names);
    res_names("mid_block.resnets.1", m.temb_names);
    for (auto& bp : m.up) for (size_t j = 0; j < bp.layers.size(); ++j) res_names("up_blocks." + std::to_string(bp.idx) + ".resnets." + std::to_string(j), m.temb_names);

    for (auto& kv : names) {
        const std::string& k = kv.first;
        const StEntry& e = *kv.second;
        std::vector<u16> t;
        if (!host_f16(f, e, t, err)) return fail("unet_create: %s", k + ": " + err);
        const auto& s = e.shape;
        HostPack hp; hp.name = k;
        if (has_suffix(k, ".weight") && s.size() == 4 && s[3] == 3) {                   // Conv2d 3x3 [O, I, 3, 3] -> OHWI, I padded to 64, O to 8
            const long long O = s[0], I = s[1], IP = (I + 63) / 64 * 64, OP = (O + 7) / 8 * 8;
            hp.rows = OP; hp.cols = 9 * IP; hp.d.assign((size_t)(OP * 9 * IP), 0);
            for (long long o = 0; o < O; ++o) for (long long i = 0; i < I; ++i) for (int y = 0; y < 3; ++y) for (int x = 0; x < 3; ++x)
                hp.d[(size_t)(((o * 3 + y) * 3 + x) * IP + i)] = t[(size_t)(((o * I + i) * 3 + y) * 3 + x)];
        } else if (has_suffix(k, ".weight") && s.size() == 5) {                         // Conv3d (3,1,1) [O, I, 3, 1, 1] -> [O, 3, I]
            const long long O = s[0], I = s[1];
            hp.rows = O; hp.cols = 3 * I; hp.d.resize((size_t)(O * 3 * I));
            for (long long o = 0; o < O; ++o) for (long long i = 0; i < I; ++i) for (int tt = 0; tt < 3; ++tt)
                hp.d[(size_t)((o * 3 + tt) * I + i)] = t[(size_t)((o * I + i) * 3 + tt)];
        } else if (has_suffix(k, "mix_factor")) {
            // AlphaBlender (resnet.py:789-802): alpha = sigmoid(mix_factor) in fp16 (`alpha.to(x_spatial.dtype)`), 1 - alpha in fp16 arithmetic
            const float mf = f16_bits_to_f32(t[0]);                                      // (load_state_dict keeps every parameter in fp16)
            const _Float16 a = (_Float16)(1.0f / (1.0f + expf(-mf)));
            const _Float16 om = (_Float16)((_Float16)1.0f - a);
            m.alpha[k] = {(double)(float)a, (double)(float)om};
            continue;
        } else {                                                                          // Linear / norm / bias / 1x1 shortcut: as stored ([rows, rest])
            hp.rows = s.empty() ? 1 : s[0];
            long long n = 1;
            for (long long d : s) n *= d;
            hp.cols = hp.rows ? n / hp.rows : 0;
            hp.d = std::move(t);
        }
        packs.push_back(std::move(hp));
    }
    // conv_out.bias padded to 8 outputs
    {
        std::vector<u16> b;
        if (!get("conv_out.bias", b)) return fail("unet_create: %s", err);
        b.resize((b.size() + 7) / 8 * 8, 0);
        for (auto& hp : packs) if (hp.name == "conv_out.bias") { hp.rows = (long long)b.size(); hp.cols = 1; hp.d = b; }
    }
    // fused qkv; GEGLU row groups; stacked time-embedding projections
    for (auto& kv : names) {
        const std::string& k = kv.first;
        if (has_suffix(k, "attn1.to_q.weight")) {
            const std::string pre = k.substr(0, k.size() - strlen("to_q.weight"));
            HostPack hp; hp.name = pre + "qkv";
            std::vector<long long> sh;
            for (const char* part : {"to_q.weight", "to_k.weight", "to_v.weight"}) {
                std::vector<u16> t;
                if (!get(pre + part, t, &sh)) return fail("unet_create: %s", err);
                hp.d.insert(hp.d.end(), t.begin(), t.end());
            }
            hp.cols = sh[1]; hp.rows = (long long)hp.d.size() / hp.cols;
            packs.push_back(std::move(hp));
        } else if (has_suffix(k, ".net.0.proj.weight")) {
            const std::string pre = k.substr(0, k.size() - strlen("weight"));
            std::vector<u16> wv, bv;
            std::vector<long long> sh;
            if (!get(k, wv, &sh) || !get(pre + "bias", bv)) return fail("unet_create: %s", err);
            const long long D2 = sh[0], K = sh[1], D = D2 / 2;
            HostPack pw, pb;
            if (K == 320) {                                                               // ops.pack_geglu_chunked: per 64-wide hidden chunk 4 x [16 hidden | 16 gate]
                if (D % 64) return fail("unet_create: %s: hidden width must be a multiple of 64", k);
                pw.name = pre + "geglu_cw"; pb.name = pre + "geglu_cb";
                pw.rows = 2 * D; pw.cols = K; pw.d.resize((size_t)(2 * D * K)); pb.rows = 2 * D; pb.cols = 1; pb.d.resize((size_t)(2 * D));
                long long r = 0;
                for (long long j = 0; j < D / 64; ++j) for (int q = 0; q < 64; q += 16) for (int half = 0; half < 2; ++half) for (int i = 0; i < 16; ++i, ++r) {
                    const long long src = (half ? D : 0) + 64 * j + q + i;
                    memcpy(&pw.d[(size_t)(r * K)], &wv[(size_t)(src * K)], (size_t)K * 2);
                    pb.d[(size_t)r] = bv[(size_t)src];
                }
            } else {                                                                      // ops.pack_geglu: per 80-column output tile [80 hidden | 80 gate], zero-padded
                const long long tiles = (D + 79) / 80;
                pw.name = pre + "geglu_w"; pb.name = pre + "geglu_b";
                pw.rows = tiles * 160; pw.cols = K; pw.d.assign((size_t)(tiles * 160 * K), 0); pb.rows = tiles * 160; pb.cols = 1; pb.d.assign((size_t)(tiles * 160), 0);
                for (long long t = 0; t < tiles; ++t) {
                    const long long n = std::min<long long>(80, D - 80 * t);
                    memcpy(&pw.d[(size_t)(160 * t * K)], &wv[(size_t)(80 * t * K)], (size_t)(n * K) * 2);
                    memcpy(&pw.d[(size_t)((160 * t + 80) * K)], &wv[(size_t)((D + 80 * t) * K)], (size_t)(n * K) * 2);
                    memcpy(&pb.d[(size_t)(160 * t)], &bv[(size_t)(80 * t)], (size_t)n * 2);
                    memcpy(&pb.d[(size_t)(160 * t + 80)], &bv[(size_t)(D + 80 * t)], (size_t)n * 2);
                }
            }
            packs.push_back(std::move(pw)); packs.push_back(std::move(pb));
            if (K != 320 && D % 128 == 0) {                                              // ops.pack_geglu64 beside it: per 64-wide hidden chunk [64 hidden | 64 gate] (k_gemm_g256, round 6)
                HostPack qw, qb;
                qw.name = pre + "geglu_w64"; qb.name = pre + "geglu_b64";
                qw.rows = 2 * D; qw.cols = K; qw.d.resize((size_t)(2 * D * K)); qb.rows = 2 * D; qb.cols = 1; qb.d.resize((size_t)(2 * D));
                for (long long j = 0; j < D / 64; ++j) {
                    memcpy(&qw.d[(size_t)(128 * j * K)], &wv[(size_t)(64 * j * K)], (size_t)(64 * K) * 2);
                    memcpy(&qw.d[(size_t)((128 * j + 64) * K)], &wv[(size_t)((D + 64 * j) * K)], (size_t)(64 * K) * 2);
                    memcpy(&qb.d[(size_t)(128 * j)], &bv[(size_t)(64 * j)], 64 * 2);
                    memcpy(&qb.d[(size_t)(128 * j + 64)], &bv[(size_t)(D + 64 * j)], 64 * 2);
                }
                packs.push_back(std::move(qw)); packs.push_back(std::move(qb));
            }
        }
    }
    {
        HostPack pw, pb; pw.name = "time_emb_proj.all.weight"; pb.name = "time_emb_proj.all.bias";
        int off = 0; long long K = 0;
        for (auto& nme : m.temb_names) {
            std::vector<u16> wv, bv;
            std::vector<long long> sh;
            if (!get(nme + ".weight", wv, &sh) || !get(nme + ".bias", bv)) return fail("unet_create: %s", err);
            K = sh[1];
            m.temb_slice[nme] = {off, off + (int)sh[0]};
            off += (int)sh[0];
            pw.d.insert(pw.d.end(), wv.begin(), wv.end());
            pb.d.insert(pb.d.end(), bv.begin(), bv.end());
        }
        pw.rows = off; pw.cols = K; pb.rows = off; pb.cols = 1;
        packs.push_back(std::move(pw)); packs.push_back(std::move(pb));
    }
    return SYN3R_OK;
}

}  // namespace

namespace {
int unet_create_impl(const char* weights_dir, const char* variant, syn3r_unet** out);
}
// The loader parses files it does not control (config.json, the safetensors header): nothing may leave through extern "C" as a
// C++ exception (bad_alloc / length_error from a hostile header would terminate a ctypes / cgo host).
extern "C" int syn3r_unet_create(const char* weights_dir, const char* variant, syn3r_unet** out) {
    SYN3R_REQUIRE(weights_dir && out, "unet_create: null argument");
    *out = nullptr;
    try {
        return unet_create_impl(weights_dir, variant, out);
    } catch (const std::exception& e) {
        set_error("unet_create: %s", e.what());
    } catch (...) {
        set_error("unet_create: unexpected exception");
    }
    *out = nullptr;
    return SYN3R_E_INVALID;
}
namespace {
int unet_create_impl(const char* weights_dir, const char* variant, syn3r_unet** out) {
    const std::string dir(weights_dir);
    std::string cfg_text;
    if (!read_file(dir + "/config.json", cfg_text)) return fail("unet_create: cannot read %s/config.json", dir);
    JParser jp{cfg_text.data(), cfg_text.data() + cfg_text.size()};
    JVal cfg = jp.value();
    if (!jp.ok || cfg.type != JVal::OBJ) return fail("unet_create: %s/config.json is not a JSON object", dir);
    std::unique_ptr<syn3r_unet> m(new syn3r_unet);
    int rc = build_plan(*m, cfg);
    if (rc) return rc;
    std::unique_ptr<StFile> fp;
    std::string err;
    auto absent = [](const std::string& e) { return e.compare(0, 11, "cannot open") == 0; };
    if (variant && *variant) {
        fp.reset(new StFile);
        if (!fp->open_(dir + "/diffusion_pytorch_model." + variant + ".safetensors", err)) {
            if (!absent(err)) return fail("unet_create: %s", err);               // the file is there and is not a checkpoint
            fp.reset();
        }
    }
    if (!fp) {
        fp.reset(new StFile);
        if (!fp->open_(dir + "/diffusion_pytorch_model.safetensors", err)) {
            if (!absent(err)) return fail("unet_create: %s", err);
            return fail("unet_create: no safetensors weights under %s (diffusion_pytorch_model[.variant].safetensors)", dir);
        }
    }
    const StFile& f = *fp;
    std::vector<HostPack> packs;
    rc = pack_weights(*m, f, packs);
    if (rc) return rc;
    rc = check_hip(hipGetDevice(&m->device), "hipGetDevice");
    if (rc) return rc;
    size_t total = 0;
    for (auto& hp : packs) total += (hp.d.size() * 2 + 255) / 256 * 256;
    rc = check_hip(hipMalloc((void**)&m->blob, total ? total : 256), "hipMalloc(unet weights)");
    if (rc) return rc;
    m->blob_bytes = total;
    size_t off = 0;
    for (auto& hp : packs) {
        rc = check_hip(hipMemcpy(m->blob + off, hp.d.data(), hp.d.size() * 2, hipMemcpyHostToDevice), "hipMemcpy(unet weights)");
        if (rc) { (void)hipFree(m->blob); return rc; }
        m->w[hp.name] = Wt{(__half*)(m->blob + off), hp.rows, hp.cols};
        off += (hp.d.size() * 2 + 255) / 256 * 256;
    }
    m->ff_ln = tune_env("SYN3R_FF_LN", 1) != 0;          // tuning builds only (common.h): the shipped library reads no environment
    m->ln_qkv = tune_env("SYN3R_LN_QKV", 1) != 0;
    *out = m.release();
    {
        std::lock_guard<std::mutex> lk(g_unet_mu);
        unet_registry().insert(*out);
    }
    return SYN3R_OK;
}
}  // namespace

extern "C" int syn3r_unet_destroy(syn3r_unet* m) {
    if (!m) return SYN3R_OK;
    {
        std::lock_guard<std::mutex> lk(g_unet_mu);
        if (!unet_registry().erase(m)) { set_error("unet_destroy: not a live handle"); return SYN3R_E_INVALID; }
    }
    for (auto& kv : m->pos_cache) {
        if (kv.second.ready) (void)hipEventDestroy(kv.second.ready);
        (void)hipFree(kv.second.p);
    }
    if (m->blob) (void)hipFree(m->blob);
    delete m;
    return SYN3R_OK;
}

// ================================================================================================ forward
namespace {

// Stream-ordered arena over the caller's workspace: first-fit with coalescing.  Every kernel of a forward runs on ONE stream,
// so a block may be handed out again as soon as the host has ENQUEUED its last reader (what a caching allocator does for
// one stream).  In `dry` mode nothing is launched and only the high-water mark is tracked: the same allocation sequence as
// the real run, so syn3r_unet_workspace_bytes is exact.
struct Arena {
    char* base = nullptr; size_t cap = 0, peak = 0;
    std::map<size_t, size_t> free_;        // offset -> size
    std::unordered_map<size_t, size_t> live;
    void reset(char* b, size_t c) { base = b; cap = c; peak = 0; free_.clear(); live.clear(); free_[0] = c; }
    void* alloc(size_t n) {
        n = (n + 255) / 256 * 256;
        if (!n) n = 256;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= n) {
                const size_t off = it->first, rest = it->second - n;
                free_.erase(it);
                if (rest) free_[off + n] = rest;
                live[off] = n;
                peak = std::max(peak, off + n);
                return base + off;
            }
        }
        return nullptr;
    }
    void release(void* p) {
        if (!p) return;
        size_t off = (size_t)((char*)p - base);
        auto it = live.find(off);
        if (it == live.end()) return;
        size_t n = it->second;
        live.erase(it);
        auto nx = free_.lower_bound(off);
        if (nx != free_.end() && off + n == nx->first) { n += nx->second; nx = free_.erase(nx); }
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { pv->second += n; return; }
        }
        free_[off] = n;
    }
};

// an activation [rows, cols] in the arena; gnp: the GroupNorm partial sums its producer was asked for (syn3r_gemm_set_gn_partials),
// gn_ok: the kernel that ran wrote them
struct T { __half* p = nullptr; long long rows = 0; int cols = 0; float* gnp = nullptr; bool gn_ok = false; };

struct Epi {                                                              // the fused epilogue of syn3r_gemm_f16 and friends
    const __half* rowvec = nullptr; long long ldrv = 0; int rows_per_vec = 0, rv_group = 0;
    const T* residual = nullptr; const T* aux = nullptr;
    float s_acc = 1.f, s_res = 1.f, s_aux = 1.f;
};

struct Run {
    syn3r_unet& m;
    Arena ar;
    bool dry = false;
    hipStream_t stream = nullptr;
    int B = 0, F = 0, h = 0, w = 0, G = 0;
    bool shared_ctx = false;
    T ehs, temb_all;
    std::unordered_map<std::string, T> cross;      // folded cross-attention vectors of this forward
    int rc = SYN3R_OK;
    explicit Run(syn3r_unet& m_) : m(m_) {}

    const Wt* W(const std::string& k) {
        auto it = m.w.find(k);
        if (it == m.w.end()) { if (!rc) { set_error("unet_forward: the checkpoint has no tensor %s", k.c_str()); rc = SYN3R_E_INVALID; } return nullptr; }
        return &it->second;
    }
    const __half* Wp(const std::string& k) { const Wt* t = W(k); return t ? t->p : nullptr; }
    bool ok() const { return rc == SYN3R_OK; }
    T make(long long rows, int cols) {
        T t; t.rows = rows; t.cols = cols;
        if (!ok()) return t;
        t.p = (__half*)ar.alloc((size_t)rows * cols * 2);
        if (!t.p) { set_error("unet_forward: workspace too small (needs syn3r_unet_workspace_bytes)"); rc = SYN3R_E_WORKSPACE; }
        return t;
    }
    void drop(T& t) { ar.release(t.p); t.p = nullptr; ar.release(t.gnp); t.gnp = nullptr; t.gn_ok = false; }
    // GroupNorm statistics from the producer's epilogue (model.py: gn_stats=TUNE["gn_epilogue"]): the buffer is part of the
    // allocation sequence whenever the shape can carry partial sums (dry runs see the same sequence); whether the kernel chosen
    // for the shape wrote them is known after the launch
    void gn_begin(T& out, bool want) {
        static const int gn_env = tune_env("SYN3R_GN_EPILOGUE", 1);
        const size_t nb = (want && gn_env != 0 && out.rows <= SYN3R_DIM_MAX) ? syn3r_gn_partials_bytes((int)out.rows, out.cols) : 0;
        if (!nb || !ok()) return;
        out.gnp = (float*)ar.alloc(nb);
        if (!out.gnp) { set_error("unet_forward: workspace too small (needs syn3r_unet_workspace_bytes)"); rc = SYN3R_E_WORKSPACE; return; }
        if (go()) chk(syn3r_gemm_set_gn_partials(out.gnp, nb));
    }
    void gn_end(T& out) {
        if (!out.gnp || !go()) return;
        out.gn_ok = syn3r_gemm_gn_partials_written() != 0;
        if (!out.gn_ok) syn3r_gemm_set_gn_partials(nullptr, 0);
    }
    void chk(int r) { if (r && !rc) rc = r; }
    bool go() const { return ok() && !dry; }

    // ---- operators (ops.py)
    T linear(const T& x, long long ldx, int K, const std::string& wname, const char* bias_name, const Epi& e = Epi(), bool gn = false) {
        const Wt* wt = W(wname);
        const __half* b = bias_name ? Wp(bias_name) : nullptr;
        T out = make(x.rows, wt ? (int)wt->rows : 0);
        gn_begin(out, gn);
        if (go())
            chk(syn3r_gemm_f16(x.p, ldx, wt->p, out.p, out.cols, b, e.rowvec, e.ldrv, e.rows_per_vec, e.rv_group,
                               e.residual ? e.residual->p : nullptr, e.residual ? e.residual->cols : 0, e.aux ? e.aux->p : nullptr,
                               e.aux ? e.aux->cols : 0, e.s_acc, e.s_res, e.s_aux, (int)x.rows, out.cols, K, stream));
        gn_end(out);
        return out;
    }
    T linear(const T& x, const std::string& wname, const std::string& bname, const Epi& e = Epi(), bool gn = false) {
        return linear(x, x.cols, x.cols, wname, bname.empty() ? nullptr : bname.c_str(), e, gn);
    }
    T linear_cat(const T& x1, const T& x2, const std::string& wname, const std::string& bname) {
        const Wt* wt = W(wname);
        const int N = wt ? (int)wt->rows : 0;
        if (ok() && syn3r_gemm_2src_supported((int)x1.rows, N, x1.cols, x2.cols, x1.cols, x2.cols)) {
            T out = make(x1.rows, N);
            if (go()) chk(syn3r_gemm_2src_f16(x1.p, x1.cols, x1.cols, x2.p, x2.cols, x2.cols, wt->p, out.p, N, Wp(bname), (int)x1.rows, N, stream));
            return out;
        }
        T cat = make(x1.rows, x1.cols + x2.cols);
        if (go()) {
            const long long n = cat.rows * cat.cols;
            hipLaunchKernelGGL(k_cat_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x1.p, x1.cols, x2.p, x2.cols, cat.p, cat.rows);
        }
        T out = linear(cat, wname, bname);
        drop(cat);
        return out;
    }
    T conv3x3(const T& x, int NB, int Hi, int Wi, int Cin, const std::string& wname, const std::string& bname, int stride, bool ups,
              const Epi& e, int* Ho_ = nullptr, int* Wo_ = nullptr, bool gn = true) {
        const Wt* wt = W(wname);
        const int Cout = wt ? (int)wt->rows : 0;
        const int Hg = ups ? 2 * Hi : Hi, Wg = ups ? 2 * Wi : Wi;
        const int Ho = (Hg + 1 - 2) / stride + 1, Wo = (Wg + 1 - 2) / stride + 1;
        if (Ho_) *Ho_ = Ho;
        if (Wo_) *Wo_ = Wo;
        T out = make((long long)NB * Ho * Wo, Cout);
        gn_begin(out, gn);
        if (go())
            chk(syn3r_conv2d3x3_f16(x.p, wt->p, out.p, Cout, Wp(bname), e.rowvec, e.ldrv, e.rows_per_vec, e.residual ? e.residual->p : nullptr,
                                    e.residual ? Cout : 0, e.s_acc, e.s_res, NB, Hi, Wi, Cin, Cout, stride, ups ? 1 : 0, 1, stream));
        gn_end(out);
        return out;
    }
    T tconv3(const T& x, const std::string& wname, const std::string& bname, int HW, const Epi& e) {
        const Wt* wt = W(wname);
        const int Cout = wt ? (int)wt->rows : 0;
        T out = make(x.rows, Cout);
        gn_begin(out, true);                     // (every temporal convolution's output is a GroupNorm input, model.py:_resblock)
        if (go())
            chk(syn3r_tconv3_f16(x.p, wt->p, out.p, Cout, Wp(bname), e.rowvec, e.ldrv, e.rows_per_vec, e.residual ? e.residual->p : nullptr,
                                 e.residual ? Cout : 0, e.s_acc, e.s_res, B, F, HW, x.cols, Cout, stream));
        gn_end(out);
        return out;
    }
    T groupnorm(const T& x, const std::string& pre, int samples, float eps, bool silu, const T* x2 = nullptr) {
        const int rows = (int)(x.rows / samples);
        const size_t wsb = syn3r_groupnorm_workspace_bytes(samples, rows);
        T out = make(x.rows, x.cols + (x2 ? x2->cols : 0));
        void* ws = ok() ? ar.alloc(wsb) : nullptr;
        if (ok() && !ws) { set_error("unet_forward: workspace too small (needs syn3r_unet_workspace_bytes)"); rc = SYN3R_E_WORKSPACE; }
        const int Ct = x.cols + (x2 ? x2->cols : 0);
        if (go() && x.gn_ok && (!x2 || x2->gn_ok) && rows % 32 == 0 && Ct % 320 == 0 && x.cols % 10 == 0) {      // ops.py:groupnorm, the same rule
            chk(syn3r_groupnorm_pre_f16(x.p, x.cols, x.gnp, x2 ? x2->p : nullptr, x2 ? x2->cols : 0, x2 ? x2->gnp : nullptr, out.p, samples, rows,
                                        Wp(pre + ".weight"), Wp(pre + ".bias"), eps, silu, ws, wsb, stream));
        } else if (go()) {
            if (x2) chk(syn3r_groupnorm_2src_f16(x.p, x.cols, x2->p, x2->cols, out.p, samples, rows, Wp(pre + ".weight"), Wp(pre + ".bias"), eps, silu, ws, wsb, stream));
            else chk(syn3r_groupnorm_f16(x.p, out.p, samples, rows, x.cols, Wp(pre + ".weight"), Wp(pre + ".bias"), eps, silu, ws, wsb, stream));
        }
        ar.release(ws);
        return out;
    }
    // y = LayerNorm(x [+ addvec]); xsum (optional) receives x + addvec
    T layernorm(const T& x, const std::string& pre, const __half* addvec = nullptr, int rows_per_vec = 0, T* xsum = nullptr) {
        T out = make(x.rows, x.cols);
        if (xsum) *xsum = make(x.rows, x.cols);
        if (go())
            chk(syn3r_layernorm_f16(x.p, out.p, xsum ? xsum->p : nullptr, addvec, rows_per_vec, x.rows, x.cols, Wp(pre + ".weight"), Wp(pre + ".bias"), 1e-5f, stream));
        return out;
    }
    // model.py:_norm_qkv - attn1's stacked q / k / v projection of norm1(x): one kernel at C = 320
    T norm_qkv(const std::string& blk, const T& x) {
        const Wt* wq = W(blk + ".attn1.qkv");
        if (m.ln_qkv && x.cols == 320 && wq && wq->rows % 320 == 0) {
            T out = make(x.rows, (int)wq->rows);
            if (go())
                chk(syn3r_layernorm_linear320_f16(x.p, x.cols, Wp(blk + ".norm1.weight"), Wp(blk + ".norm1.bias"), 1e-5f, wq->p, out.p, out.cols,
                                                  (int)x.rows, (int)wq->rows, x.cols, stream));
            return out;
        }
        T n1 = layernorm(x, blk + ".norm1");
        T qkv = linear(n1, blk + ".attn1.qkv", "");
        drop(n1);
        return qkv;
    }
    T attention(const T& qkv, int nseq, int S, int heads) {
        const int C = heads * 64;
        T out = make(qkv.rows, C);
        if (go()) chk(syn3r_attention_f16(qkv.p, qkv.p + C, qkv.p + 2 * C, 3 * C, out.p, C, nseq, S, heads, stream));
        return out;
    }
    T attention_temporal(const T& qkv, int HW, int heads) {
        const int C = heads * 64;
        T out = make(qkv.rows, C);
        if (go()) chk(syn3r_attention_temporal_f16(qkv.p, qkv.p + C, qkv.p + 2 * C, 3 * C, out.p, C, B, F, HW, heads, stream));
        return out;
    }
    // model.py:_ff - FeedForward of block `pre` on x, `norm` = the LayerNorm in front of it (fused at C = 320)
    T ff(const std::string& pre, const T& x_in, const std::string& norm, const Epi& e) {
        const Wt* w0 = W(pre + ".net.0.proj.weight");
        const int D = w0 ? (int)(w0->rows / 2) : 0;
        const std::string w2 = pre + ".net.2.weight", b2 = pre + ".net.2.bias";
        T x = x_in;
        bool own = false;
        T out;
        if (m.w.count(pre + ".net.0.proj.geglu_cw")) {
            const bool fuse_ln = !norm.empty() && m.ff_ln;
            if (!norm.empty() && !fuse_ln) { x = layernorm(x_in, norm); own = true; }
            out = make(x.rows, x.cols);
            if (go()) {
                const void *r = e.residual ? e.residual->p : nullptr, *a = e.aux ? e.aux->p : nullptr;
                const long long ldr = e.residual ? e.residual->cols : 0, lda = e.aux ? e.aux->cols : 0;
                if (fuse_ln)
                    chk(syn3r_feedforward_fused_ln_f16(x.p, x.cols, Wp(norm + ".weight"), Wp(norm + ".bias"), 1e-5f, Wp(pre + ".net.0.proj.geglu_cw"),
                                                       Wp(pre + ".net.0.proj.geglu_cb"), D, Wp(w2), Wp(b2), out.p, out.cols, r, ldr, a, lda,
                                                       e.s_acc, e.s_res, e.s_aux, (int)x.rows, x.cols, stream));
                else
                    chk(syn3r_feedforward_fused_f16(x.p, x.cols, Wp(pre + ".net.0.proj.geglu_cw"), Wp(pre + ".net.0.proj.geglu_cb"), D, Wp(w2), Wp(b2),
                                                    out.p, out.cols, r, ldr, a, lda, e.s_acc, e.s_res, e.s_aux, (int)x.rows, x.cols, stream));
            }
        } else {
            if (!norm.empty()) { x = layernorm(x_in, norm); own = true; }
            const Wt* wt2 = W(w2);
            const int N = wt2 ? (int)wt2->rows : 0;
            out = make(x.rows, N);
            const size_t wsb = syn3r_feedforward_workspace_bytes((int)x.rows, D);
            void* ws = ok() ? ar.alloc(wsb) : nullptr;
            if (ok() && !ws) { set_error("unet_forward: workspace too small (needs syn3r_unet_workspace_bytes)"); rc = SYN3R_E_WORKSPACE; }
            // model.py:_ff - net.0 on the 256 x 256 tile where the shape has whole tiles (bit-identical to the 80-column path)
            static const int g256_env = tune_env("SYN3R_FF_G256", 1);
            const bool p64 = g256_env != 0 && m.w.count(pre + ".net.0.proj.geglu_w64") && syn3r_feedforward_p64_supported((int)x.rows, D, x.cols);
            if (go() && p64)
                chk(syn3r_feedforward_p64_f16(x.p, x.cols, Wp(pre + ".net.0.proj.geglu_w64"), Wp(pre + ".net.0.proj.geglu_b64"), D, Wp(w2), Wp(b2), out.p, N,
                                              e.residual ? e.residual->p : nullptr, e.residual ? e.residual->cols : 0, e.aux ? e.aux->p : nullptr,
                                              e.aux ? e.aux->cols : 0, e.s_acc, e.s_res, e.s_aux, (int)x.rows, x.cols, N, ws, wsb, stream));
            else if (go())
                chk(syn3r_feedforward_f16(x.p, x.cols, Wp(pre + ".net.0.proj.geglu_w"), Wp(pre + ".net.0.proj.geglu_b"), D, Wp(w2), Wp(b2), out.p, N,
                                          e.residual ? e.residual->p : nullptr, e.residual ? e.residual->cols : 0, e.aux ? e.aux->p : nullptr,
                                          e.aux ? e.aux->cols : 0, e.s_acc, e.s_res, e.s_aux, (int)x.rows, x.cols, N, ws, wsb, stream));
            ar.release(ws);
        }
        if (own) drop(x);
        return out;
    }
    T silu(const T& x) {
        T y = make(x.rows, x.cols);
        const long long n = x.rows * x.cols;
        if (go()) hipLaunchKernelGGL(k_silu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x.p, y.p, n);
        return y;
    }
    // attn2 with a single key: to_out(to_v(ctx)) per context row (model.py:_cross_vec); kept for the forward
    const T& cross_vec(const std::string& pre) {
        auto it = cross.find(pre);
        if (it != cross.end()) return it->second;
        T v = linear(ehs, pre + ".to_v.weight", "");
        T o = linear(v, pre + ".to_out.0.weight", pre + ".to_out.0.bias");
        drop(v);
        return cross.emplace(pre, o).first->second;
    }

    // SpatioTemporalResBlock (resnet.py:325-378, 613-636, 789-802); x2: the block input is [x | x2] (up blocks)
    T resblock(const std::string& pre, const T& x, int cin, int cout, const T* x2) {
        const int HW = h * w;
        const std::string s = pre + ".spatial_res_block", t = pre + ".temporal_res_block";
        auto slice = [&](const std::string& name, Epi& e) {
            auto it = m.temb_slice.find(name);
            e.rowvec = temb_all.p ? temb_all.p + (it != m.temb_slice.end() ? it->second.first : 0) : nullptr;
            if (dry) e.rowvec = nullptr;
            e.ldrv = temb_all.cols; e.rows_per_vec = F * HW;
        };
        T h1 = groupnorm(x, s + ".norm1", B * F, 1e-5f, true, x2);
        Epi e1; slice(s + ".time_emb_proj", e1);
        T h2 = conv3x3(h1, B * F, h, w, cin, s + ".conv1.weight", s + ".conv1.bias", 1, false, e1);
        drop(h1);
        T h3 = groupnorm(h2, s + ".norm2", B * F, 1e-5f, true);
        drop(h2);
        T skip = x;
        bool own_skip = false;
        if (x2) { skip = linear_cat(x, *x2, s + ".conv_shortcut.weight", s + ".conv_shortcut.bias"); own_skip = true; }
        else if (cin != cout) { skip = linear(x, s + ".conv_shortcut.weight", s + ".conv_shortcut.bias"); own_skip = true; }
        Epi e2; e2.residual = &skip;
        T xs = conv3x3(h3, B * F, h, w, cout, s + ".conv2.weight", s + ".conv2.bias", 1, false, e2);
        drop(h3);
        if (own_skip) drop(skip);
        T g1 = groupnorm(xs, t + ".norm1", B, 1e-5f, true);
        Epi e3; slice(t + ".time_emb_proj", e3);
        T c1 = tconv3(g1, t + ".conv1.weight", t + ".conv1.bias", HW, e3);
        drop(g1);
        T g2 = groupnorm(c1, t + ".norm2", B, 1e-5f, true);
        drop(c1);
        auto al = m.alpha.find(pre + ".time_mixer.mix_factor");
        const double a = al != m.alpha.end() ? al->second.first : 0.5, om = al != m.alpha.end() ? al->second.second : 0.5;
        if (al == m.alpha.end() && !rc) { set_error("unet_forward: the checkpoint has no %s.time_mixer.mix_factor", pre.c_str()); rc = SYN3R_E_INVALID; }
        Epi e4; e4.residual = &xs; e4.s_acc = (float)om; e4.s_res = (float)(a + om);     // alpha xs + (1 - alpha)(xs + conv2)
        T out = tconv3(g2, t + ".conv2.weight", t + ".conv2.bias", HW, e4);
        drop(g2);
        drop(xs);
        return out;
    }

    // frame-position embedding of block `pre` (transformer_temporal.py:326-337), one row per (b, f): a function of the weights, F, B
    const __half* pos_embedding(const std::string& pre, int ch) {
        const std::string key = pre + "|" + std::to_string(F) + "|" + std::to_string(B);
        auto it = m.pos_cache.find(key);
        if (it != m.pos_cache.end() && !dry) {
            // filled on the stream of the forward that created it: a forward on ANOTHER stream orders itself behind that fill
            // (the header's "nothing synchronises" holds for the host; this is a device-side dependency)
            if (it->second.stream != stream && it->second.ready) chk(check_hip(hipStreamWaitEvent(stream, it->second.ready, 0), "hipStreamWaitEvent(position embedding)"));
            return it->second.p;
        }
        T pos = make(F, ch);
        if (go()) hipLaunchKernelGGL(k_timestep_embedding, dim3((unsigned)((F * (ch / 2) + 255) / 256)), dim3(256), 0, stream, (const float*)nullptr, 0.0, 1.0f, F, ch, pos.p);
        T e1 = linear(pos, pre + ".time_pos_embed.linear_1.weight", pre + ".time_pos_embed.linear_1.bias");
        drop(pos);
        T e1s = silu(e1);
        drop(e1);
        T e2 = linear(e1s, pre + ".time_pos_embed.linear_2.weight", pre + ".time_pos_embed.linear_2.bias");
        drop(e1s);
        __half* keep = nullptr;
        if (go()) {
            chk(check_hip(hipMalloc((void**)&keep, (size_t)B * F * ch * 2), "hipMalloc(position embedding)"));
            if (ok()) {
                const long long n = (long long)B * F * ch;
                hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, e2.p, keep, F, (long long)B * F, ch);
                syn3r_unet::PosEmb pe;
                pe.p = keep; pe.stream = stream;
                if (hipEventCreateWithFlags(&pe.ready, hipEventDisableTiming) == hipSuccess) {
                    if (hipEventRecord(pe.ready, stream) != hipSuccess) { (void)hipEventDestroy(pe.ready); pe.ready = nullptr; }
                } else pe.ready = nullptr;
                m.pos_cache[key] = pe;
            } else if (keep) { (void)hipFree(keep); keep = nullptr; }
        }
        drop(e2);
        return keep;
    }

    // TransformerSpatioTemporalModel (transformer_temporal.py:277-379) with its two blocks (attention.py:283-533)
    T transformer(const std::string& pre, const T& x, int ch, int heads) {
        const int HW = h * w;
        T gn = groupnorm(x, pre + ".norm", B * F, 1e-6f, false);
        T hs = linear(gn, pre + ".proj_in.weight", pre + ".proj_in.bias");
        drop(gn);
        const __half* emb = pos_embedding(pre, ch);
        const std::string b = pre + ".transformer_blocks.0", t = pre + ".temporal_transformer_blocks.0";
        {   // BasicTransformerBlock
            T qkv = norm_qkv(b, hs);
            T a1 = attention(qkv, B * F, HW, heads);
            drop(qkv);
            const T& cv = cross_vec(b + ".attn2");
            Epi e; e.residual = &hs; e.rowvec = dry ? nullptr : cv.p; e.ldrv = cv.cols; e.rows_per_vec = (shared_ctx ? B : 1) * F * HW;
            T o = linear(a1, b + ".attn1.to_out.0.weight", b + ".attn1.to_out.0.bias", e);
            drop(a1);
            drop(hs);
            hs = o;
            Epi ef; ef.residual = &hs;
            T f2 = ff(b + ".ff", hs, b + ".norm3", ef);
            drop(hs);
            hs = f2;
        }
        // TemporalBasicTransformerBlock on hs + emb
        T tt;
        if (m.ff_ln && m.w.count(t + ".ff_in.net.0.proj.geglu_cw")) {          // C = 320: add, norm_in, ff_in and + residual in one kernel
            tt = make(hs.rows, hs.cols);
            const Wt* w0 = W(t + ".ff_in.net.0.proj.weight");
            if (go())
                chk(syn3r_feedforward_fused_addln_f16(hs.p, hs.cols, emb, HW, Wp(t + ".norm_in.weight"), Wp(t + ".norm_in.bias"), 1e-5f,
                                                      Wp(t + ".ff_in.net.0.proj.geglu_cw"), Wp(t + ".ff_in.net.0.proj.geglu_cb"), w0 ? (int)(w0->rows / 2) : 0,
                                                      Wp(t + ".ff_in.net.2.weight"), Wp(t + ".ff_in.net.2.bias"), tt.p, tt.cols, nullptr, 0, 1.f, 1.f, 0.f,
                                                      (int)hs.rows, hs.cols, stream));
        } else {
            T hmix;
            T nin = layernorm(hs, t + ".norm_in", emb, HW, &hmix);
            Epi e0; e0.residual = &hmix;
            tt = ff(t + ".ff_in", nin, "", e0);
            drop(nin);
            drop(hmix);
        }
        {
            T qkv = norm_qkv(t, tt);
            T a1 = attention_temporal(qkv, HW, heads);
            drop(qkv);
            // the reference's batch-interleaved temporal context (transformer_temporal.py:310-317 vs attention.py:487-489): see model.py
            int rpv, grp;
            if (shared_ctx) { rpv = B * F * HW; grp = 0; }
            else if (G == 1) { rpv = F * HW; grp = 0; }
            else {
                if (HW % G && !rc) { set_error("unet_forward: the temporal cross-attention context interleave needs h*w divisible by the group size %d", G); rc = SYN3R_E_INVALID; }
                rpv = -G; grp = G != B ? G * F * HW : 0;
            }
            const T& cv = cross_vec(t + ".attn2");
            Epi e; e.residual = &tt; e.rowvec = dry ? nullptr : cv.p; e.ldrv = cv.cols; e.rows_per_vec = rpv; e.rv_group = grp;
            T o = linear(a1, t + ".attn1.to_out.0.weight", t + ".attn1.to_out.0.bias", e);
            drop(a1);
            drop(tt);
            tt = o;
        }
        auto al = m.alpha.find(pre + ".time_mixer.mix_factor");
        const double a = al != m.alpha.end() ? al->second.first : 0.5, om = al != m.alpha.end() ? al->second.second : 0.5;
        if (al == m.alpha.end() && !rc) { set_error("unet_forward: the checkpoint has no %s.time_mixer.mix_factor", pre.c_str()); rc = SYN3R_E_INVALID; }
        Epi em; em.residual = &tt; em.aux = &hs; em.s_acc = (float)om; em.s_res = (float)om; em.s_aux = (float)a;   // alpha hs + (1 - alpha)(ff + tt)
        T mix = ff(t + ".ff", tt, t + ".norm3", em);
        drop(tt);
        drop(hs);
        Epi eo; eo.residual = &x;
        T out = linear(mix, pre + ".proj_out.weight", pre + ".proj_out.bias", eo, true);
        drop(mix);
        return out;
    }

    // unet_spatio_temporal_condition.py:356-489
    void forward(const __half* sample, double timestep, const __half* ehs_in, int ehs_rows, const float* added_ids, __half* out) {
        const int c0 = m.boc[0];
        // split-K scratch for the lowest level's convolutions (model.py:forward, the same rule): set for this forward, on this thread
        void* splitk = nullptr;
        {
            const int down = 1 << (m.boc.size() - 1);
            const long long m_low = (long long)B * F * (h / down) * (w / down);
            if (tune_env("SYN3R_SPLITK", 1) != 0 && ((m_low + 255) / 256) * ((m.boc.back() + 159) / 160) * 2 <= 256) {
                const size_t bytes = (size_t)4 * m_low * m.boc.back() * 4;
                splitk = ok() ? ar.alloc(bytes) : nullptr;
                if (ok() && !splitk) { set_error("unet_forward: workspace too small (needs syn3r_unet_workspace_bytes)"); rc = SYN3R_E_WORKSPACE; }
                if (go()) chk(syn3r_gemm_set_splitk_workspace(splitk, bytes));
            }
        }
        forward_body(sample, timestep, ehs_in, ehs_rows, added_ids, out);
        if (splitk) {
            if (!dry) syn3r_gemm_set_splitk_workspace(nullptr, 0);
            ar.release(splitk);
        }
    }
    void forward_body(const __half* sample, double timestep, const __half* ehs_in, int ehs_rows, const float* added_ids, __half* out) {
        const int c0 = m.boc[0];
        // 1. time (:385-418)
        T te = make(B, c0);
        if (go()) hipLaunchKernelGGL(k_timestep_embedding, dim3((unsigned)((B * (c0 / 2) + 255) / 256)), dim3(256), 0, stream, (const float*)nullptr, timestep, 0.0f, B, c0, te.p);
        T e1 = linear(te, "time_embedding.linear_1.weight", "time_embedding.linear_1.bias");
        drop(te);
        T e1s = silu(e1);
        drop(e1);
        T emb = linear(e1s, "time_embedding.linear_2.weight", "time_embedding.linear_2.bias");
        drop(e1s);
        const int nid = m.proj_in / m.add_dim;                             // time ids per sample (3: fps, motion bucket, noise aug)
        T ta = make((long long)B * nid, m.add_dim);
        if (go()) hipLaunchKernelGGL(k_timestep_embedding, dim3((unsigned)((B * nid * (m.add_dim / 2) + 255) / 256)), dim3(256), 0, stream, added_ids, 0.0, 0.0f, B * nid, m.add_dim, ta.p);
        T ta2 = ta; ta2.rows = B; ta2.cols = nid * m.add_dim;              // reshape(B, -1)
        T a1 = linear(ta2, "add_embedding.linear_1.weight", "add_embedding.linear_1.bias");
        drop(ta);
        T a1s = silu(a1);
        drop(a1);
        T aug = linear(a1s, "add_embedding.linear_2.weight", "add_embedding.linear_2.bias");
        drop(a1s);
        T es = make(emb.rows, emb.cols);
        if (go()) { const long long n = es.rows * es.cols; hipLaunchKernelGGL(k_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, emb.p, aug.p, es.p, n); }
        drop(emb);
        drop(aug);
        T act = silu(es);
        drop(es);
        temb_all = linear(act, "time_emb_proj.all.weight", "time_emb_proj.all.bias");
        drop(act);
        ehs.p = (__half*)ehs_in; ehs.rows = ehs_rows; ehs.cols = m.cross[0];
        // 2. conv_in on NHWC with the channels padded to 64 (:428)
        const int CP = (m.in_ch + 63) / 64 * 64;
        T xin = make((long long)B * F * h * w, CP);
        if (go()) { const long long n = xin.rows * CP; hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sample, xin.p, B * F, m.in_ch, h * w, CP); }
        T x = conv3x3(xin, B * F, h, w, CP, "conv_in.weight", "conv_in.bias", 1, false, Epi());
        drop(xin);
        std::vector<T> skips{x};
        // 3. down (:432-449)
        for (auto& bp : m.down) {
            const std::string bpre = "down_blocks." + std::to_string(bp.idx);
            for (size_t j = 0; j < bp.layers.size(); ++j) {
                T y = resblock(bpre + ".resnets." + std::to_string(j), x, bp.layers[j].first, bp.layers[j].second, nullptr);
                if (bp.attn) { T z = transformer(bpre + ".attentions." + std::to_string(j), y, bp.layers[j].second, bp.heads); drop(y); y = z; }
                x = y;                                                     // (the previous x lives on in `skips`)
                skips.push_back(x);
            }
            if (bp.resample) {
                int Ho, Wo;
                T y = conv3x3(x, B * F, h, w, bp.ch, bpre + ".downsamplers.0.conv.weight", bpre + ".downsamplers.0.conv.bias", 2, false, Epi(), &Ho, &Wo);
                h = Ho; w = Wo;
                x = y;
                skips.push_back(x);
            }
        }
        // 4. mid (:452-457)
        {
            const int mid = m.boc.back();
            T y = resblock("mid_block.resnets.0", x, mid, mid, nullptr);      // x = the last skip: stays
            T z = transformer("mid_block.attentions.0", y, mid, m.heads.back());
            drop(y);
            x = resblock("mid_block.resnets.1", z, mid, mid, nullptr);
            drop(z);
        }
        // 5. up (:460-478): the skip is read in place (norm1 and the shortcut projection take two sources)
        for (auto& bp : m.up) {
            const std::string bpre = "up_blocks." + std::to_string(bp.idx);
            for (size_t j = 0; j < bp.layers.size(); ++j) {
                T sk = skips.back();
                skips.pop_back();
                T y = resblock(bpre + ".resnets." + std::to_string(j), x, bp.layers[j].first, bp.layers[j].second, &sk);
                drop(x);
                drop(sk);
                if (bp.attn) { T z = transformer(bpre + ".attentions." + std::to_string(j), y, bp.layers[j].second, bp.heads); drop(y); y = z; }
                x = y;
            }
            if (bp.resample) {
                int Ho, Wo;
                T y = conv3x3(x, B * F, h, w, bp.ch, bpre + ".upsamplers.0.conv.weight", bpre + ".upsamplers.0.conv.bias", 1, true, Epi(), &Ho, &Wo);
                drop(x);
                h = Ho; w = Wo;
                x = y;
            }
        }
        // 6. out (:481-486)
        T gn = groupnorm(x, "conv_norm_out", B * F, 1e-5f, true);
        drop(x);
        T y = conv3x3(gn, B * F, h, w, c0, "conv_out.weight", "conv_out.bias", 1, false, Epi(), nullptr, nullptr, false);
        drop(gn);
        if (go()) { const long long n = (long long)B * F * m.out_ch * h * w; hipLaunchKernelGGL(k_nhwc_to_nchw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, y.p, out, B * F, m.out_ch, h * w, y.cols); }
        drop(y);
        for (auto& kv : cross) drop(kv.second);
        cross.clear();
        drop(temb_all);
    }
};

int check_shape(const syn3r_unet* m, int B, int F, int h, int w, int ehs_rows, int ctx_group) {
    SYN3R_REQUIRE(unet_live(m), "unet: not a live handle (syn3r_unet_create)");
    SYN3R_REQUIRE(B > 0 && B <= 64 && F > 0 && F <= 32 && SYN3R_SIDE_OK(h) && SYN3R_SIDE_OK(w), "unet: bad sizes B=%d F=%d h=%d w=%d (F <= 32)", B, F, h, w);
    SYN3R_REQUIRE((long long)B * F * h * w <= SYN3R_DIM_MAX, "unet: B*F*h*w = %lld rows is beyond the operators' limit", (long long)B * F * h * w);
    const int down = 1 << (m->boc.size() - 1);
    SYN3R_REQUIRE(h % down == 0 && w % down == 0, "unet: h and w must be multiples of %d", down);
    SYN3R_REQUIRE(ehs_rows == 1 || ehs_rows == B, "unet: encoder_hidden_states must have 1 (shared) or B rows, got %d", ehs_rows);
    SYN3R_REQUIRE(ctx_group >= 0 && (ctx_group == 0 || B % ctx_group == 0), "unet: ctx_group=%d must divide the batch size %d", ctx_group, B);
    return SYN3R_OK;
}

}  // namespace

extern "C" size_t syn3r_unet_workspace_bytes(syn3r_unet* m, int B, int F, int h, int w, int ehs_rows) {
    if (check_shape(m, B, F, h, w, ehs_rows, 0)) return 0;
    try {
    Run r(*m);
    r.dry = true;
    r.ar.reset((char*)4096, (size_t)1 << 46);
    r.B = B; r.F = F; r.h = h; r.w = w; r.G = 1;                    // (the context grouping changes epilogue arguments, not buffers)
    r.shared_ctx = B == 1 || ehs_rows == 1;
    r.forward(nullptr, 0.0, (const __half*)4096, r.shared_ctx ? 1 : B, nullptr, nullptr);
    return r.ok() ? r.ar.peak + 256 : 0;
    } catch (...) { set_error("unet_workspace_bytes: unexpected exception"); return 0; }
}

extern "C" int syn3r_unet_forward(syn3r_unet* m, const void* sample, double timestep, const void* encoder_hidden_states, int ehs_rows,
                                  const float* added_time_ids, void* out, int B, int F, int h, int w, int ctx_group, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    int rc = check_shape(m, B, F, h, w, ehs_rows, ctx_group);
    if (rc) return rc;
    SYN3R_REQUIRE(sample && encoder_hidden_states && added_time_ids && out, "unet_forward: null tensor");
    SYN3R_REQUIRE(workspace && ((uintptr_t)workspace % 256) == 0, "unet_forward: workspace must be non-null and 256-byte aligned");
    SYN3R_REQUIRE((((uintptr_t)sample | (uintptr_t)encoder_hidden_states | (uintptr_t)out) % 16) == 0, "unet_forward: misaligned tensor");
    try {
    Run r(*m);
    r.ar.reset((char*)workspace, workspace_bytes / 256 * 256);
    r.stream = (hipStream_t)stream;
    r.B = B; r.F = F; r.h = h; r.w = w;
    r.G = ctx_group > 0 ? ctx_group : B;
    r.shared_ctx = B == 1 || ehs_rows == 1;
    r.forward((const __half*)sample, timestep, (const __half*)encoder_hidden_states, r.shared_ctx ? 1 : B, added_time_ids, (__half*)out);
    if (r.ok()) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return check_hip(e, "unet_forward launch");
    }
    return r.rc;
    } catch (const std::exception& e) { set_error("unet_forward: %s", e.what()); return SYN3R_E_INVALID;
    } catch (...) { set_error("unet_forward: unexpected exception"); return SYN3R_E_INVALID; }
}
