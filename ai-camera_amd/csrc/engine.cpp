// engine.cpp -- loads an .aicw engine file and runs it: the replacement for TRTEngine
// (src/trt_utils/trt_engine.py:15-216), plus the fused detector / ReID entry points that replace
// YOLODetector.detect (src/detector/yolo_detector.py:68-149) and the crop->embed half of
// DeepSORT.update (src/tracker/deepsort_tracker.py:104-113, src/tracker/reid_model.py:67-126).
//
// Data layout in HBM: every activation is NHWC ([items][h][w][c], c contiguous) in the activation
// dtype (fp16 or fp32); concatenations are channel slices of one buffer; head outputs and
// embeddings are fp32.  Weights are repacked once at load to [cout][kh][kw][cin] with K padded to
// the MFMA K-step and zero rows up to cout_pad, so the conv kernel needs no bounds checks on them.
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>

namespace aic {

namespace {

struct Reader {
    const char* p;
    size_t n, off = 0;
    template <class T> const T* take(size_t count) {
        AIC_REQUIRE(off + count * sizeof(T) <= n, AIC_ERR_FORMAT, "engine file truncated");
        const T* r = reinterpret_cast<const T*>(p + off);
        off += count * sizeof(T);
        return r;
    }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// IEEE fp32 -> fp16 bits, round-to-nearest-even (host side; no dependence on compiler-rt helpers)
inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            // rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                          // rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                   // subnormal half
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7FFFFFu; }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return (uint16_t)(sign | (base + q));
}

}  // namespace

Model::Model(Device& d, const void* blob, size_t nbytes, int dtype_, int max_items_)
    : dev(&d), dtype(dtype_), max_items(max_items_) {
    AIC_REQUIRE(dtype == AIC_F16 || dtype == AIC_F32, AIC_ERR_INVALID, "dtype must be AIC_F16 or AIC_F32");
    AIC_REQUIRE(max_items > 0, AIC_ERR_INVALID, "max_items must be positive");
    d.use();
    blob_copy = std::make_shared<const std::vector<char>>(reinterpret_cast<const char*>(blob), reinterpret_cast<const char*>(blob) + nbytes);
    Reader rd{reinterpret_cast<const char*>(blob), nbytes};
    const uint32_t* head = rd.take<uint32_t>(2);
    AIC_REQUIRE(head[0] == 0x57434941u && head[1] == 1u, AIC_ERR_FORMAT, "not an AICW v1 engine file");
    const int32_t* h = rd.take<int32_t>(15);
    kind = h[0], in_h = h[1], in_w = h[2];
    const int nb = h[3], no = h[4], nw = h[5], nout = h[6];
    for (int i = 0; i < 8; ++i) meta[i] = h[7 + i];
    AIC_REQUIRE((kind == KIND_YOLO || kind == KIND_REID) && nb > 0 && no > 0 && nw >= 0 && nout > 0, AIC_ERR_FORMAT,
                "bad engine header");
    const int32_t* btab = rd.take<int32_t>((size_t)nb * 4);
    const int32_t* otab = rd.take<int32_t>((size_t)no * 20);
    const int64_t* wtab = rd.take<int64_t>((size_t)nw * 6);
    const int32_t* outtab = rd.take<int32_t>((size_t)nout * 8);
    const size_t payload_off = rd.off;
    const float* payload = reinterpret_cast<const float*>(rd.p + payload_off);
    const size_t payload_n = (nbytes - payload_off) / 4;

    const int esz = dtype == AIC_F16 ? 2 : 4;
    const int vec = 16 / esz;
    // ---- activation arena
    bufs.resize(nb);
    storage.resize(nb);
    for (int i = 0; i < nb; ++i) {
        BufDesc& b = bufs[i];
        b.h = btab[i * 4], b.w = btab[i * 4 + 1], b.c = btab[i * 4 + 2], b.f32 = btab[i * 4 + 3];
        AIC_REQUIRE(b.h > 0 && b.w > 0 && b.c > 0, AIC_ERR_FORMAT, "bad buffer shape");
        b.esize = b.f32 ? 4 : esz;
        AIC_REQUIRE(b.c % (16 / b.esize) == 0, AIC_ERR_FORMAT, "buffer channel count must be a multiple of 16 bytes");
        b.per_item = (size_t)b.h * b.w * b.c * b.esize;
        storage[i].alloc(b.per_item * max_items + 256);
        HIP_CHECK(hipMemsetAsync(storage[i].p, 0, storage[i].n, d.s_main));
        b.p = storage[i].p;
    }
    AIC_REQUIRE(bufs[0].h == in_h && bufs[0].w == in_w && bufs[0].c == 8, AIC_ERR_FORMAT, "input buffer must be HxWx8");
    d_zero.alloc(256);
    HIP_CHECK(hipMemsetAsync(d_zero.p, 0, 256, d.s_main));
    // ---- ops
    ops.resize(no);
    for (int i = 0; i < no; ++i) std::memcpy(ops[i].v, otab + (size_t)i * 20, 80);
    outs.resize(nout);
    for (int i = 0; i < nout; ++i) std::memcpy(outs[i].v, outtab + (size_t)i * 8, 32);
    // ---- weights: OIHW fp32 -> [cout_pad][Kp] (kh, kw, cin) in the activation dtype
    weights.resize(nw);
    const int bke = 4 * vec;
    // one device weight = the output channels of one or more engine-file weights stacked (a merged conv, below)
    struct WSrc { const float* w; const float* b; int cout; };
    // `second` (one part only): a 1x1 conv of cin2 channels with the same output channels, folded in as K columns behind the window's
    // (ConvArgs::x2); the bias becomes the sum of the two
    auto pack_weights = [&](ConvWeights& w, const std::vector<WSrc>& parts, int cin, int kh, int kw, const WSrc* second = nullptr, int cin2 = 0) {
        w.cout = 0;
        for (const WSrc& p : parts) w.cout += p.cout;
        w.cin = cin, w.kh = kh, w.kw = kw;
        w.cin_eff = w.cin == 3 ? 8 : w.cin;
        AIC_REQUIRE(w.cin_eff % vec == 0, AIC_ERR_FORMAT, "conv input channels must be a multiple of 16 bytes");
        w.cin2 = second ? cin2 : 0;
        w.K = w.kh * w.kw * w.cin_eff + w.cin2;
        w.Kp = round_up(w.K, bke);
        w.cout_pad = round_up(w.cout, 128) + 128;
        // + 8 K-steps of zero slack: the conv kernel's drain iterations step the weight pointer past the last row
        std::vector<char> packed(((size_t)w.cout_pad * w.Kp + 8 * bke) * esz, 0);
        std::vector<float> bias(w.cout_pad, 0.f);
        int co0 = 0;
        for (const WSrc& p : parts) {
            for (int co = 0; co < p.cout; ++co)
                for (int ci = 0; ci < w.cin; ++ci)
                    for (int ky = 0; ky < w.kh; ++ky)
                        for (int kx = 0; kx < w.kw; ++kx) {
                            const float v = p.w[(((size_t)co * w.cin + ci) * w.kh + ky) * w.kw + kx];
                            const size_t k = (size_t)(co0 + co) * w.Kp + (size_t)(ky * w.kw + kx) * w.cin_eff + ci;
                            if (dtype == AIC_F16) reinterpret_cast<uint16_t*>(packed.data())[k] = f32_to_f16_bits(v);
                            else reinterpret_cast<float*>(packed.data())[k] = v;
                        }
            std::copy(p.b, p.b + p.cout, bias.begin() + co0);
            co0 += p.cout;
        }
        if (second) {
            const size_t k0 = (size_t)w.kh * w.kw * w.cin_eff;
            for (int co = 0; co < second->cout; ++co) {
                for (int ci = 0; ci < cin2; ++ci) {
                    const float v = second->w[(size_t)co * cin2 + ci];
                    const size_t k = (size_t)co * w.Kp + k0 + ci;
                    if (dtype == AIC_F16) reinterpret_cast<uint16_t*>(packed.data())[k] = f32_to_f16_bits(v);
                    else reinterpret_cast<float*>(packed.data())[k] = v;
                }
                bias[co] += second->b[co];
            }
        }
        w.w.alloc(packed.size());
        HIP_CHECK(hipMemcpy(w.w.p, packed.data(), packed.size(), hipMemcpyHostToDevice));
        w.bias.alloc(w.cout_pad);
        HIP_CHECK(hipMemcpy(w.bias.p, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    };
    auto wsrc_of = [&](int i) {
        const int cout = (int)wtab[i * 6], cin = (int)wtab[i * 6 + 1], kh = (int)wtab[i * 6 + 2], kw = (int)wtab[i * 6 + 3];
        const size_t woff = (size_t)wtab[i * 6 + 4], boff = (size_t)wtab[i * 6 + 5];
        AIC_REQUIRE(woff + (size_t)cout * cin * kh * kw <= payload_n && boff + cout <= payload_n, AIC_ERR_FORMAT, "weight offsets out of range");
        return WSrc{payload + woff, payload + boff, cout};
    };
    for (int i = 0; i < nw; ++i)
        pack_weights(weights[i], {wsrc_of(i)}, (int)wtab[i * 6 + 1], (int)wtab[i * 6 + 2], (int)wtab[i * 6 + 3]);
    // ---- validate ops against buffers, count FLOPs
    for (const OpDesc& o : ops) {
        const int* v = o.v;
        AIC_REQUIRE(v[1] >= 0 && v[1] < nb && v[4] >= 0 && v[4] < nb, AIC_ERR_FORMAT, "op references a missing buffer");
        const BufDesc& sb = bufs[v[1]];
        const BufDesc& db = bufs[v[4]];
        const int cin_chk = (v[0] == OP_CONV && v[3] == 3) ? 8 : v[3];
        AIC_REQUIRE(v[2] >= 0 && v[2] + cin_chk <= sb.c && v[2] % vec == 0, AIC_ERR_FORMAT, "op source slice out of range");
        if (v[0] == OP_CONV) {
            AIC_REQUIRE(v[15] >= 0 && v[15] < nw, AIC_ERR_FORMAT, "conv references a missing weight");
            const ConvWeights& w = weights[v[15]];
            AIC_REQUIRE(w.cout == v[6] && w.cin == v[3] && w.kh == v[7] && w.kw == v[8], AIC_ERR_FORMAT, "conv/weight shape mismatch");
            AIC_REQUIRE(v[5] >= 0 && v[5] + v[6] <= db.c && v[5] % 4 == 0, AIC_ERR_FORMAT, "conv destination slice out of range");
            AIC_REQUIRE((sb.h + 2 * v[10] - v[7]) / v[9] + 1 == db.h && (sb.w + 2 * v[10] - v[8]) / v[9] + 1 == db.w,
                        AIC_ERR_FORMAT, "conv geometry mismatch");
            AIC_REQUIRE(!sb.f32 || dtype == AIC_F32, AIC_ERR_FORMAT, "conv input must be an activation buffer");
            if (v[14]) {
                AIC_REQUIRE(v[12] >= 0 && v[12] < nb, AIC_ERR_FORMAT, "residual references a missing buffer");
                const BufDesc& rb = bufs[v[12]];
                AIC_REQUIRE(rb.h == db.h && rb.w == db.w && v[13] + v[6] <= rb.c && !rb.f32, AIC_ERR_FORMAT, "residual shape mismatch");
            }
            flops_per_item += 2.0 * db.h * db.w * (double)v[6] * v[3] * v[7] * v[8];
            ++n_convs;
        }
    }
    // ---- fold: a ResNet downsample branch (1x1 / stride s, linear, read only as the residual of the block's last conv) into that conv:
    // relu(conv3x3(t) + b + ds(x) + b') is ONE GEMM over K = [window of t | channels of x] (ConvArgs::x2).  The 1x1's launch goes away and
    // with it its output tensor -- written once, read once, 2 GB per 15 360 crops on ReID layer2 (1.58 ms of downsample launches per
    // 512-frame group, and the 9 us residual epilogue of the three convs that added them).  The sum no longer passes through an fp16
    // rounding of the branch: not bit-identical to the unfolded graph, 8e-5 from it and as far from the fp32 oracle as it is
    // (tests/test_gpu_nets.py::test_downsample_branch_folded_into_last_conv).
    // AICAM_NO_DS_FOLD=1: off.
    if (dtype == AIC_F16 && !getenv("AICAM_NO_DS_FOLD") && !getenv("AICAM_NO_FUSE")) {
        for (size_t j = 0; j < ops.size(); ++j) {
            int* c = ops[j].v;
            if (c[0] != OP_CONV || c[14] != 1 || ops[j].fuse || c[15] >= nw || c[13] != 0 || c[16] != 0) continue;
            const int rbuf = c[12];
            size_t i = ops.size();
            for (size_t k = 0; k < j; ++k)
                if (ops[k].v[0] == OP_CONV && ops[k].v[4] == rbuf && !ops[k].fuse) i = k;
            if (i == ops.size()) continue;
            const int* p = ops[i].v;
            if (p[7] != 1 || p[8] != 1 || p[10] != 0 || p[11] != 0 || p[14] != 0 || p[5] != 0 || p[6] != c[6] || p[15] >= nw) continue;
            if (bufs[rbuf].c != c[6] || bufs[rbuf].f32) continue;
            bool clash = false;                         // nobody else reads or writes the branch's tensor; its source is still intact at the last conv
            for (size_t k = 0; k < ops.size(); ++k) {
                const int* u = ops[k].v;
                if (k != i && u[4] == rbuf) clash = true;
                if (u[1] == rbuf || (k != j && u[0] == OP_CONV && u[14] && u[12] == rbuf)) clash = true;
                if (k > i && k <= j && u[4] == p[1]) clash = true;
            }
            for (auto& o : outs)
                if (o.v[0] == rbuf) clash = true;
            if (clash) continue;
            const BufDesc& xb = bufs[c[1]];
            const BufDesc& yb = bufs[c[4]];
            ConvArgs q{};
            q.H = xb.h, q.W = xb.w, q.Cin = c[3], q.Ho = yb.h, q.Wo = yb.w, q.Cout = c[6], q.KH = c[7], q.KW = c[8], q.stride = c[9], q.pad = c[10];
            q.out_f32 = yb.f32;
            if (!conv_x2_supported(dtype, q, p[3])) continue;
            const WSrc second = wsrc_of(p[15]);
            weights.emplace_back();
            pack_weights(weights.back(), {wsrc_of(c[15])}, c[3], c[7], c[8], &second, p[3]);
            c[15] = (int)weights.size() - 1;
            c[14] = 0, c[12] = 0, c[13] = 0;            // no residual any more
            c[16] = p[1] + 1, c[17] = p[2], c[18] = p[3], c[19] = p[9];
            ops[i].fuse = 2;                            // absorbed: run_range skips it
            storage[rbuf].release();
            bufs[rbuf].p = nullptr;
        }
    }
    // ---- fold: a 2x nearest-neighbour upsample whose output slice is read by exactly one 1x1 conv, as the FIRST channels of its input
    // (YOLOv8's neck: up(P5) | P4 -> 12.c2f.cv1, up(12) | P3 -> 15.c2f.cv1), into that conv: it reads those channels from the
    // half-resolution tensor at (y >> 1, x >> 1) (ConvArgs::xs).  Same values in the same K order: bit-identical
    // (tests/test_gpu_nets.py::test_upsample_folded_into_its_reader).  AICAM_NO_UP_FOLD=1: off.
    if (!getenv("AICAM_NO_UP_FOLD") && !getenv("AICAM_NO_FUSE")) {
        auto overlap = [](int a0, int an, int b0, int bn) { return a0 < b0 + bn && b0 < a0 + an; };
        for (size_t i = 0; i < ops.size(); ++i) {
            const int* u = ops[i].v;
            if (u[0] != OP_UPSAMPLE2X || ops[i].fuse) continue;
            const int S = u[1], sc = u[2], c = u[3], D = u[4], dc = u[5];
            size_t j = ops.size();
            bool clash = false;
            for (size_t k = 0; k < ops.size(); ++k) {
                if (k == i || ops[k].fuse == 2) continue;
                const int* w = ops[k].v;
                const bool reads = w[1] == D && overlap(w[2], w[0] == OP_CONV && w[3] == 3 ? 8 : w[3], dc, c);
                if (reads) {
                    if (j != ops.size() || k < i) clash = true;
                    j = k;
                }
                if (w[0] == OP_CONV && w[14] && w[12] == D && overlap(w[13], w[6], dc, c)) clash = true;
                if (w[4] == D && overlap(w[5], w[6], dc, c)) clash = true;                      // another writer of the slice
            }
            for (auto& o : outs)
                if (o.v[0] == D || (kind == KIND_YOLO && o.v[1] == D)) clash = true;
            if (clash || j == ops.size()) continue;
            int* v = ops[j].v;
            if (v[0] != OP_CONV || ops[j].fuse || v[2] != dc || v[3] <= c || v[16] != 0 || v[15] >= (int)weights.size()) continue;
            for (size_t k = i + 1; k <= j; ++k)                                                  // the low-resolution source is still intact at the reader
                if (ops[k].fuse != 2 && ops[k].v[4] == S && overlap(ops[k].v[5], ops[k].v[6], sc, c)) clash = true;
            if (clash) continue;
            const BufDesc& xb = bufs[v[1]];
            const BufDesc& yb = bufs[v[4]];
            const ConvWeights& w = weights[v[15]];
            ConvArgs q{};
            q.H = xb.h, q.W = xb.w, q.Cin = w.cin_eff, q.Ho = yb.h, q.Wo = yb.w, q.Cout = v[6], q.KH = v[7], q.KW = v[8], q.stride = v[9], q.pad = v[10];
            q.Kp = w.Kp, q.out_f32 = yb.f32;
            if (2 * bufs[S].h != xb.h || 2 * bufs[S].w != xb.w || !conv_xs_supported(dtype, q, c)) continue;
            // (a conv that takes the next 1x1 into its epilogue -- fuse 3, marked below -- is launched with a tail; the marking skips
            // every conv with a split source, and conv_tail_supported() refuses one as well)
            ops[j].xs_buf = S, ops[j].xs_coff = sc, ops[j].xs_c = c;
            ops[i].fuse = 2;                            // absorbed: run_range skips it
        }
    }
    // ---- merge: two convs that read the SAME tensor slice with the same window, stride, padding and activation and no residual
    // (YOLOv8's detect branches: 22.box{l}.0 and 22.cls{l}.0 both start from the level's feature map) become ONE conv whose output
    // channels are the two sets side by side in one new buffer; their readers take channel slices of it.  The map is read once
    // instead of twice and the GEMM is 144 wide instead of 64 and 80 (levels with maps up to 40 x 40, see below).  Per output channel nothing changes (same K order, same
    // epilogue): the head is bit-identical (tests/test_gpu_nets.py::test_merged_detect_branch_heads).  AICAM_NO_MERGE=1: off.
    if (dtype == AIC_F16 && !getenv("AICAM_NO_MERGE") && !getenv("AICAM_NO_FUSE")) {
        for (size_t i = 0; i < ops.size(); ++i) {
            int* c = ops[i].v;
            if (c[0] != OP_CONV || c[14] != 0 || ops[i].fuse || c[15] >= nw) continue;
            for (size_t j = i + 1; j < ops.size() && j <= i + 8; ++j) {
                int* p = ops[j].v;
                if (p[0] != OP_CONV || p[14] != 0 || ops[j].fuse || p[15] >= nw) continue;
                bool same = p[1] == c[1] && p[2] == c[2] && p[3] == c[3];
                for (int k = 7; k <= 11; ++k) same = same && p[k] == c[k];
                const int di = c[4], dj = p[4];
                if (!same || di == dj || c[5] != 0 || p[5] != 0 || c[7] != 3) continue;
                const BufDesc bi = bufs[di], bj = bufs[dj];
                if (bi.f32 || bj.f32 || bi.c != c[6] || bj.c != p[6] || bi.h != bj.h || bi.w != bj.w || (c[6] % vec) || (p[6] % vec)) continue;
                // maps up to 40 x 40 only (measured, profiles/r03: on the 80 x 80 level the 64-channel conv has the 3x3 patch kernel at
                // 645 TFLOP/s and the 144-wide 4-wave tile reaches 470: 1 157 us merged against 633 + 374; on the 40 x 40 and 20 x 20
                // levels the merged conv wins, 490 against 513 us and 233 against 286)
                static const int merge_px = [] { const char* e = getenv("AICAM_MERGE_MAXPX"); return e ? atoi(e) : 1600; }();
                if (bi.h * bi.w > merge_px) continue;
                bool clash = false;                     // the second conv now runs at the first one's place: nobody may touch its output in between,
                for (size_t k = 0; k < ops.size(); ++k) {   // and nobody else may write either buffer
                    const int* u = ops[k].v;
                    if (k != i && k != j && (u[4] == di || u[4] == dj)) clash = true;
                    if (k > i && k < j && (u[1] == dj || (u[0] == OP_CONV && u[14] && u[12] == dj))) clash = true;
                }
                for (auto& o : outs)
                    if (o.v[0] == di || o.v[0] == dj || (kind == KIND_YOLO && (o.v[1] == di || o.v[1] == dj))) clash = true;
                if (clash) continue;
                // new buffer and weight
                BufDesc nb = bi;
                nb.c = c[6] + p[6];
                nb.per_item = (size_t)nb.h * nb.w * nb.c * nb.esize;
                storage.emplace_back();
                storage.back().alloc(nb.per_item * max_items + 256);
                HIP_CHECK(hipMemsetAsync(storage.back().p, 0, storage.back().n, d.s_main));
                nb.p = storage.back().p;
                bufs.push_back(nb);
                const int m = (int)bufs.size() - 1;
                weights.emplace_back();
                pack_weights(weights.back(), {wsrc_of(c[15]), wsrc_of(p[15])}, c[3], c[7], c[8]);
                const int ci_out = c[6];
                for (auto& o : ops) {                   // readers of either output: slices of the merged buffer
                    int* u = o.v;
                    if (u[1] == di) u[1] = m;
                    else if (u[1] == dj) { u[1] = m; u[2] += ci_out; }
                    if (u[0] == OP_CONV && u[14]) {
                        if (u[12] == di) u[12] = m;
                        else if (u[12] == dj) { u[12] = m; u[13] += ci_out; }
                    }
                }
                c[4] = m, c[5] = 0, c[6] = nb.c, c[15] = (int)weights.size() - 1;
                ops[j].fuse = 2;                        // absorbed: run_range skips it
                storage[di].release(), storage[dj].release();      // (nobody reads the two old buffers any more)
                bufs[di].p = bufs[dj].p = nullptr;
                break;
            }
        }
    }
    // ---- fusion: conv3x3/1 (3->64)+ReLU followed by max-pool 3x3/2 of exactly that tensor (ReID stem)
    if (dtype == AIC_F16 && !getenv("AICAM_NO_FUSE")) {
        for (size_t i = 0; i + 1 < ops.size(); ++i) {
            const int* c = ops[i].v;
            const int* p = ops[i + 1].v;
            if (c[0] != OP_CONV || p[0] != OP_MAXPOOL3S2) continue;
            const BufDesc& cb = bufs[c[4]];
            const bool conv_ok = c[3] == 3 && c[6] == 64 && c[7] == 3 && c[8] == 3 && c[9] == 1 && c[10] == 1 && c[11] == 2 &&
                                 c[14] == 0 && c[5] == 0 && cb.c == 64 && !cb.f32 && cb.w == 64 && cb.h % 8 == 0 && c[2] == 0;
            const bool pool_ok = p[1] == c[4] && p[2] == 0 && p[3] == 64 && !bufs[p[4]].f32;
            bool other_reader = false;
            for (size_t j = 0; j < ops.size(); ++j)
                if (j != i + 1 && (ops[j].v[1] == c[4] || (ops[j].v[0] == OP_CONV && ops[j].v[14] && ops[j].v[12] == c[4]))) other_reader = true;
            if (conv_ok && pool_ok && !other_reader) { ops[i].fuse = 1; ops[i + 1].fuse = 2; }
        }
    }
    // ---- fusion: a SiLU conv with 64 or 80 output channels whose output feeds exactly one 1x1/1/0 conv and nothing else (YOLOv8's
    // detect branches: 22.box*.1 -> .2, 22.cls*.1 -> .2).  The 1x1 runs in the first conv's epilogue on the tile in registers;
    // the intermediate tensor is never written.  Marked here (graph properties); run_range() asks conv_tail_supported() for the rest.
    if (dtype == AIC_F16 && !getenv("AICAM_NO_FUSE")) {
        for (size_t i = 0; i + 1 < ops.size(); ++i) {
            const int* c = ops[i].v;
            const int* p = ops[i + 1].v;
            if (c[0] != OP_CONV || p[0] != OP_CONV || ops[i].fuse || ops[i + 1].fuse) continue;
            // a lead with a split source (folded upsample) or a second source (folded downsample) has no tail form: the tail kernels would
            // ignore that source and read channels nobody wrote
            if (ops[i].xs_buf >= 0 || c[16] != 0) continue;
            const bool lead_ok = (c[6] == 64 || c[6] == 80) && c[11] == 1 && c[14] == 0 && !bufs[c[4]].f32;
            const bool tail_ok = p[7] == 1 && p[8] == 1 && p[9] == 1 && p[10] == 0 && p[14] == 0 && p[1] == c[4] && p[2] == c[5] &&
                                 p[3] == c[6] && p[6] <= c[6];
            if (!lead_ok || !tail_ok) continue;
            bool other_reader = false;
            for (size_t j = 0; j < ops.size(); ++j)
                if (j != i + 1 && (ops[j].v[1] == c[4] || (ops[j].v[0] == OP_CONV && ops[j].v[14] && ops[j].v[12] == c[4]))) other_reader = true;
            for (auto& o : outs)
                if (o.v[0] == c[4] || (kind == KIND_YOLO && o.v[1] == c[4])) other_reader = true;
            if (!other_reader) ops[i].fuse = 3;          // the follower keeps fuse = 0: it runs on its own whenever the lead cannot take it
        }
    }
    {   // sub-batching plan: the maximal prefix of ops whose outputs are >= min_kb per item
        const char* e_items = getenv("AICAM_SB_ITEMS");
        const char* e_kb = getenv("AICAM_SB_MINKB");
        sub_items = e_items ? atoi(e_items) : 0;   // off by default: measured no gain on MI355X (profiles/)
        const size_t min_bytes = (size_t)(e_kb ? atoi(e_kb) : 200) * 1024;
        lead_ops = 0;
        if (sub_items > 0) {
            while (lead_ops < ops.size()) {
                const OpDesc& o = ops[lead_ops];
                const BufDesc& db = bufs[o.fuse == 1 ? ops[lead_ops + 1].v[4] : o.v[4]];
                if (o.fuse != 2 && db.per_item < min_bytes) break;
                ++lead_ops;
            }
        }
    }
    if (kind == KIND_YOLO) {
        AIC_REQUIRE(nout <= 4, AIC_ERR_FORMAT, "at most 4 detection levels");
        n_anchors = 0;
        for (auto& o : outs) n_anchors += o.v[3] * o.v[4];
        out_dim = meta[0];
        max_det_cap = 1024;
        const size_t ba = (size_t)max_items * n_anchors;
        d_boxes.alloc(ba * 4), d_maxlogit.alloc(ba), d_labels.alloc(ba);
        d_ncand.alloc(max_items), d_numdets.alloc(max_items);
        d_out_boxes.alloc((size_t)max_items * max_det_cap * 4), d_out_boxes_orig.alloc((size_t)max_items * max_det_cap * 4);
        d_out_scores.alloc((size_t)max_items * max_det_cap), d_out_labels.alloc((size_t)max_items * max_det_cap);
    } else {
        out_dim = outs[0].v[1];
        AIC_REQUIRE(bufs[outs[0].v[0]].f32, AIC_ERR_FORMAT, "embedding buffer must be fp32");
    }
    plan_side_heads();
    HIP_CHECK(hipStreamSynchronize(d.s_main));
}

Model::~Model() {
    for (int i = 0; i < 2; ++i) {
        if (side_fork[i]) (void)hipEventDestroy(side_fork[i]);
        if (side_join[i]) (void)hipEventDestroy(side_join[i]);
        if (side_stream[i] && side_stream[i] != dev->s_det && side_stream[i] != dev->s_reid) (void)hipStreamDestroy(side_stream[i]);
    }
}

// Which ops may leave the main stream (engine.hpp, side_heads).  An op's slices: what it reads (source, residual, second source, split
// source) and the one it writes, as (buffer, first channel, channels); elementwise ops count with their whole buffers.  Level l's set =
// the live ops from which output l and no other output is reachable through those slices; it has to be ONE run of consecutive live ops,
// its fork point `cut` = one past the last op outside the set that writes something the set reads, and nothing that can run beside it
// (any op at or behind `cut` outside the set) may write what the set reads or writes, or read what it writes.  Anything else: no plan.
void Model::plan_side_heads() {
    side_heads.clear();
    static const int opt = [] { const char* e = getenv("AICAM_SIDE_HEADS"); return e ? atoi(e) : 2; }();
    side_max_items = opt;
    if (kind != KIND_YOLO || opt <= 0 || outs.size() < 2 || (lead_ops > 0 && sub_items > 0)) return;
    struct Slice { int buf, c0, cn; };
    const size_t no = ops.size();
    std::vector<std::vector<Slice>> rd(no);
    std::vector<Slice> wr(no);
    auto whole = [&](int b) { return Slice{b, 0, 1 << 30}; };
    for (size_t i = 0; i < no; ++i) {
        const OpDesc& o = ops[i];
        const int* v = o.v;
        if (o.fuse == 2) continue;
        if (o.fuse == 1) return;                          // (a fused stem + pool: not a detector)
        if (v[0] == OP_CONV) {
            rd[i].push_back(Slice{v[1], v[2], v[3] == 3 ? 8 : v[3]});
            if (v[14]) rd[i].push_back(Slice{v[12], v[13], v[6]});
            if (v[16]) rd[i].push_back(Slice{v[16] - 1, v[17], v[18]});
            if (o.xs_buf >= 0) rd[i].push_back(Slice{o.xs_buf, o.xs_coff, o.xs_c});
            wr[i] = Slice{v[4], v[5], v[6]};
        } else {
            rd[i].push_back(whole(v[1]));
            wr[i] = whole(v[4]);
        }
    }
    auto hit = [](const Slice& a, const Slice& b) { return a.buf == b.buf && a.c0 < b.c0 + b.cn && b.c0 < a.c0 + a.cn; };
    auto live = [&](size_t i) { return ops[i].fuse != 2; };
    // levels reachable from each op (backwards over the list: a later reader of what an op writes)
    std::vector<unsigned> reach(no, 0u);
    for (size_t ii = no; ii-- > 0;) {
        if (!live(ii)) continue;
        for (size_t l = 0; l < outs.size(); ++l)
            if (wr[ii].buf == outs[l].v[0] || wr[ii].buf == outs[l].v[1]) reach[ii] |= 1u << l;
        for (size_t j = ii + 1; j < no; ++j) {
            if (!live(j)) continue;
            for (const Slice& r : rd[j])
                if (hit(r, wr[ii])) reach[ii] |= reach[j];
        }
    }
    for (size_t l = 0; l + 1 < outs.size() && side_heads.size() < 2; ++l) {
        size_t a = no, b = 0;
        for (size_t i = 0; i < no; ++i)
            if (live(i) && reach[i] == (1u << l)) { a = std::min(a, i); b = std::max(b, i + 1); }
        if (a >= b) continue;
        bool ok = true;
        for (size_t i = a; i < b; ++i)
            if (live(i) && reach[i] != (1u << l)) ok = false;             // one run of consecutive live ops
        size_t cut = 0;
        for (size_t i = a; i < b && ok; ++i) {
            if (!live(i)) continue;
            for (size_t k = 0; k < a; ++k) {
                if (!live(k)) continue;
                for (const Slice& r : rd[i])
                    if (hit(r, wr[k])) cut = std::max(cut, k + 1);
            }
        }
        if (!ok || cut == 0 || cut > a) continue;
        if (ops[cut - 1].fuse == 3) continue;                             // (never between a lead and the 1x1 it may take as its tail)
        for (size_t k = cut; k < no && ok; ++k) {                         // whoever may run beside the set
            if (!live(k) || (k >= a && k < b)) continue;
            for (size_t i = a; i < b && ok; ++i) {
                if (!live(i)) continue;
                if (hit(wr[k], wr[i])) ok = false;
                for (const Slice& r : rd[i])
                    if (hit(r, wr[k])) ok = false;
                for (const Slice& r : rd[k])
                    if (hit(r, wr[i])) ok = false;
            }
        }
        if (!ok) continue;
        if (!side_heads.empty() && (cut < side_heads.back().cut || a < side_heads.back().b || cut > side_heads[0].a)) continue;   // forks and sets in list order, every fork in front of every set
        side_heads.push_back(SideHead{a, b, cut});
    }
    if (getenv("AICAM_SIDE_DBG"))
        for (const SideHead& h : side_heads) fprintf(stderr, "[aicam] side head: ops [%zu, %zu) of %zu fork behind op %zu\n", h.a, h.b, no, h.cut);
    for (size_t i = 0; i < side_heads.size(); ++i) {
        side_stream[i] = getenv("AICAM_SIDE_OWN") ? nullptr : (i == 0 ? dev->s_det : dev->s_reid);
        if (!side_stream[i]) HIP_CHECK(hipStreamCreateWithFlags(&side_stream[i], hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&side_fork[i], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&side_join[i], hipEventDisableTiming));
    }
}

void Model::run_ops(size_t op0, int n, hipStream_t s) {
    const bool prof_conv = (dev->prof_mask >> PROF_CONV) & 1u;            // the HIP-event brackets describe one stream: measured runs stay on it
    if (!side_ok || side_heads.empty() || n > side_max_items || prof_conv || n_items_dev || op0 > side_heads[0].cut || s != dev->s_main) {
        run_range(op0, ops.size(), 0, n, s);
        return;
    }
    size_t at = op0;
    try {
    for (size_t i = 0; i < side_heads.size(); ++i) {                      // main stream up to each fork, the level's ops behind it on their stream
        const SideHead& h = side_heads[i];
        run_range(at, h.cut, 0, n, s);
        at = h.cut;
        HIP_CHECK(hipEventRecord(side_fork[i], s));
        HIP_CHECK(hipStreamWaitEvent(side_stream[i], side_fork[i], 0));
        run_range(h.a, h.b, 0, n, side_stream[i]);
        HIP_CHECK(hipEventRecord(side_join[i], side_stream[i]));
    }
    for (size_t i = 0; i <= side_heads.size(); ++i) {                     // the rest of the list, around the sets
        const size_t e = i < side_heads.size() ? side_heads[i].a : ops.size();
        if (at < e) run_range(at, e, 0, n, s);
        if (i < side_heads.size()) at = std::max(at, side_heads[i].b);
    }
    for (size_t i = 0; i < side_heads.size(); ++i) HIP_CHECK(hipStreamWaitEvent(s, side_join[i], 0));
    } catch (...) {                                                       // a launch failed between a fork and its join: nothing of this call may still run when the next one starts
        for (size_t i = 0; i < side_heads.size(); ++i) (void)hipStreamSynchronize(side_stream[i]);
        (void)hipStreamSynchronize(s);
        throw;
    }
}

bool Model::input_pix4_ok() const {
    static const bool off = getenv("AICAM_NO_PIX4") != nullptr;
    return !off && kind == KIND_REID && dtype == AIC_F16 && !ops.empty() && ops[0].fuse == 1 && !(lead_ops > 0 && sub_items > 0) &&
           reid_stem2_usable(in_h, in_w);
}

void Model::run(int n, hipStream_t s) {
    AIC_REQUIRE(n >= 0 && n <= max_items, AIC_ERR_CAPACITY, "batch exceeds the engine's max_items");
    if (n == 0) return;
    cls_reduced = box_decoded = 0;
    AIC_REQUIRE(!n_items_dev || !(lead_ops > 0 && sub_items > 0), AIC_ERR_INVALID, "a device-side item count cannot be combined with sub-batching");
    if (lead_ops > 0 && sub_items > 0 && n > sub_items + sub_items / 2) {
        // producer -> consumer tensors of the first layers exceed the 256 MiB Infinity Cache at full batch:
        // walk them in sub-batches so each layer reads what the previous one just wrote from cache, not HBM
        for (int i0 = 0; i0 < n; i0 += sub_items) run_range(0, lead_ops, i0, std::min(sub_items, n - i0), s);
        run_range(lead_ops, ops.size(), 0, n, s);
    } else {
        run_ops(0, n, s);
    }
}

void Model::run_frames(const uint8_t* frames, int n, const LetterboxGeom& g, hipStream_t s) {
    AIC_REQUIRE(n >= 0 && n <= max_items, AIC_ERR_CAPACITY, "batch exceeds the engine's max_items");
    if (n == 0) return;
    static const bool no_fuse = getenv("AICAM_NO_FUSE_LB") != nullptr;
    const int* v = ops[0].v;
    const bool stem = kind == KIND_YOLO && dtype == AIC_F16 && !no_fuse && ops[0].fuse == 0 && v[0] == OP_CONV && v[1] == 0 && v[2] == 0 &&
                      v[3] == 3 && v[6] == 16 && v[7] == 3 && v[8] == 3 && v[9] == 2 && v[10] == 1 && v[11] == 1 && v[14] == 0 &&
                      !(lead_ops > 0 && sub_items > 0);
    if (stem) {
        const ConvWeights& w = weights[v[15]];
        const BufDesc& db = bufs[v[4]];
        bool ok;
        {
            Prof pr(*dev, PROF_LETTERBOX, s, 2.0 * n * db.h * db.w * 16.0 * 27.0,
                    (double)n * ((double)g.src_h * g.src_w * 3 + (double)db.h * db.w * 32.0));
            ok = launch_yolo_stem_fused(frames, n, g, w.w.p, w.bias.p, w.Kp, db.p, db.c, v[5], db.h, db.w, s);
        }
        if (ok) {
            cls_reduced = box_decoded = 0;
            run_ops(1, n, s);
            return;
        }
    }
    {
        Prof pr(*dev, PROF_LETTERBOX, s, 0, (double)n * ((double)g.src_h * g.src_w * 3 + 16.0 * in_h * in_w));
        launch_letterbox(frames, n, g, 1, dtype, input(), s);
    }
    run(n, s);
}

void Model::run_range(size_t op0, size_t op1, int i0, int n, hipStream_t s) {
    auto at = [&](const BufDesc& b) { return static_cast<char*>(b.p) + (size_t)i0 * b.per_item; };
    // HIP-event timing of the conv kernel brackets RUNS of consecutive conv launches (one event pair per
    // run, not per launch: two event records per launch cost 13 % of end-to-end throughput); the summed
    // time therefore includes the ~1-2 us dependent-launch gaps inside a run (a conservative `achieved`).
    const bool prof_conv = (dev->prof_mask >> PROF_CONV) & 1u;
    bool span_open = false;
    auto span_close = [&] { if (span_open) { dev->prof_end(PROF_CONV, s); span_open = false; } };
    for (size_t oi = op0; oi < op1; ++oi) {
        const OpDesc& o = ops[oi];
        const int* v = o.v;
        BufDesc sb = bufs[v[1]];
        BufDesc db = bufs[v[4]];
        sb.p = at(sb), db.p = at(db);
        if (o.fuse == 2) continue;
        if (o.fuse == 1 || v[0] != OP_CONV) span_close();
        if (o.fuse == 1) {
            const ConvWeights& w = weights[v[15]];
            const int* pv = ops[oi + 1].v;
            BufDesc pb = bufs[pv[4]];
            pb.p = at(pb);
            const double fl = 2.0 * n * sb.h * sb.w * 64.0 * 27.0;
            Prof pr(*dev, PROF_CONV_DIRECT, s, fl, (double)n * (sb.h * sb.w * 16.0 + pb.h * pb.w * 128.0));
            launch_reid_stem_pool(sb.p, w.w.p, w.bias.p, pb.p, n, sb.h, sb.w, w.Kp, pb.c, pv[5], in_pix4 ? 4 : 8, s,
                                  (crop_src.frames && in_pix4 && i0 == 0) ? &crop_src : nullptr, n_items_dev);
            continue;
        }
        if (v[0] == OP_CONV) {
            double fl = 0, by = 0;
            auto conv_args = [&](size_t k) {
                const int* u = ops[k].v;
                BufDesc xb = bufs[u[1]], yb = bufs[u[4]];
                xb.p = at(xb), yb.p = at(yb);
                const ConvWeights& w = weights[u[15]];
                ConvArgs a{};
                a.x = xb.p, a.w = w.w.p, a.bias = w.bias.p, a.y = yb.p;
                a.x_cs = xb.c, a.x_coff = u[2], a.H = xb.h, a.W = xb.w, a.Cin = w.cin_eff;
                a.y_cs = yb.c, a.y_coff = u[5], a.Ho = yb.h, a.Wo = yb.w, a.Cout = w.cout;
                a.res = nullptr, a.r_cs = 0, a.r_coff = 0, a.res_mode = u[14], a.act = u[11];
                if (u[14]) { a.res = at(bufs[u[12]]), a.r_cs = bufs[u[12]].c, a.r_coff = u[13]; }
                a.KH = w.kh, a.KW = w.kw, a.stride = u[9], a.pad = u[10];
                if (ops[k].xs_buf >= 0) {               // folded 2x upsample: the first channels come from the half-resolution tensor
                    const BufDesc& bs = bufs[ops[k].xs_buf];
                    a.xs = at(bs), a.xs_cs = bs.c, a.xs_coff = ops[k].xs_coff, a.Hs = bs.h, a.Ws = bs.w, a.Cs = ops[k].xs_c;
                }
                if (u[16]) {                            // folded 1x1 second source
                    const BufDesc& b2 = bufs[u[16] - 1];
                    a.x2 = at(b2), a.x2_cs = b2.c, a.x2_coff = u[17], a.H2 = b2.h, a.W2 = b2.w, a.s2 = u[19], a.Cin2 = u[18];
                    fl += 2.0 * n * yb.h * yb.w * (double)w.cout * u[18];
                    by += ((double)n * yb.h * yb.w * u[18] + (double)w.cout * u[18]) * (dtype == AIC_F16 ? 2 : 4);
                }
                a.Kp = w.Kp, a.M = n * yb.h * yb.w, a.out_f32 = yb.f32, a.cout_pad = w.cout_pad, a.zero = d_zero.p;
                a.tap_rows = 0;
                a.n_dev = n_items_dev;
                for (int kh = 0; kh < w.kh; ++kh) a.tap_rows |= 1u << (kh * w.kw);
                AIC_REQUIRE(w.kh * w.kw <= 25, AIC_ERR_FORMAT, "kernel window larger than 5x5");
                fl += 2.0 * a.M * (double)w.cout * w.cin * w.kh * w.kw;
                by += ((double)n * xb.h * xb.w * w.cin + (double)a.M * w.cout) * (dtype == AIC_F16 ? 2 : 4) +
                      (double)w.cout * w.cin * w.kh * w.kw * (dtype == AIC_F16 ? 2 : 4);
                return a;
            };
            const ConvArgs a = conv_args(oi);
            // a 64-channel BasicBlock (this conv and the next, which adds this one's input) runs as ONE kernel where it applies;
            // FLOPs and algorithmic bytes are accounted as for the two convs
            bool pair = false;
            ConvArgs a2{};
            if (dtype == AIC_F16 && oi + 1 < op1 && ops[oi + 1].v[0] == OP_CONV && ops[oi + 1].fuse == 0 && a.Cin == 64 && a.Cout == 64 &&
                a.res_mode == 0 && ops[oi + 1].v[14] == 1) {
                const double fl0 = fl, by0 = by;
                a2 = conv_args(oi + 1);
                pair = a2.res == a.x && a2.x == a.y;
                if (!pair) fl = fl0, by = by0;
            }
            if (prof_conv && !span_open) { dev->prof_begin(PROF_CONV, s, 0, 0); span_open = true; }
            // a C2f block with 16-channel halves (YOLOv8n's 160 x 160 stage: cv1, m.cv1, m.cv2 + shortcut, cv2) runs as ONE kernel where
            // it applies; FLOPs and algorithmic bytes are accounted as for the four convs
            if (!pair && dtype == AIC_F16 && oi + 3 < op1 && a.KH == 1 && a.Cin == 32 && a.Cout == 32 && ops[oi + 1].v[0] == OP_CONV &&
                ops[oi + 2].v[0] == OP_CONV && ops[oi + 3].v[0] == OP_CONV && !ops[oi + 1].fuse && !ops[oi + 2].fuse && !ops[oi + 3].fuse) {
                const double fl0 = fl, by0 = by;
                const ConvArgs b1 = conv_args(oi + 1), b2 = conv_args(oi + 2), b3 = conv_args(oi + 3);
                if (conv_try_c2f16(a, b1, b2, b3, s)) {
                    if (prof_conv) dev->prof_account(PROF_CONV, fl, by);
                    oi += 3;
                    continue;
                }
                fl = fl0, by = by0;
            }
            if (pair && conv_try_c64_block(a, a2, s)) {
                if (prof_conv) dev->prof_account(PROF_CONV, fl, by);
                ++oi;
                continue;
            }
            if (pair) { fl = 0, by = 0; (void)conv_args(oi); }          // not fused after all: account this conv alone
            // a conv whose only consumer is the 1x1 conv after it (marked at load time) takes that conv into its epilogue where the
            // kernels can; FLOPs and algorithmic bytes are accounted as for the two convs
            if (o.fuse == 3 && oi + 1 < op1) {
                const double fl0 = fl, by0 = by;
                const ConvArgs t = conv_args(oi + 1);
                if (conv_tail_supported(dtype, a, t)) {
                    ConvArgs at = a;
                    at.w_tail = t.w, at.b_tail = t.bias, at.y_tail = t.y, at.t_cout = t.Cout, at.t_kp = t.Kp;
                    at.t_y_cs = t.y_cs, at.t_y_coff = t.y_coff, at.t_out_f32 = t.out_f32, at.t_act = t.act;
                    static const bool cls_reduce_on = getenv("AICAM_NO_CLS_REDUCE") == nullptr;
                    static const bool box_decode_on = getenv("AICAM_NO_BOX_DECODE") == nullptr;
                    if (reduce_cls && kind == KIND_YOLO && t.out_f32 && t.act == 0 && t.y_coff == 0 && t.y_cs == t.Cout) {
                        // a detect level's class / box branch: its logits only feed decode's arg-max / DFL expectation
                        const bool is_cls = cls_reduce_on && t.Cout == meta[0], is_box = box_decode_on && meta[1] == 16 && t.Cout == 64 && a.Cout == 64 && !is_cls;   // tail_1x1 decodes in place only in its NT == 4 form (lead Cout 64; YOLOv8x leads with 80)
                        int a0 = 0;
                        for (size_t l = 0; l < outs.size() && (is_cls || is_box); ++l) {
                            const int* ov = outs[l].v;
                            const BufDesc& ob = bufs[ov[is_cls ? 1 : 0]];
                            if (static_cast<char*>(ob.p) + (size_t)i0 * ob.per_item == static_cast<char*>(t.y) && ov[3] * ov[4] == t.Ho * t.Wo) {
                                at.t_hw = ov[3] * ov[4], at.t_a0 = a0, at.t_na = n_anchors;
                                if (is_cls) {
                                    at.t_max = d_maxlogit.p + (size_t)i0 * n_anchors, at.t_arg = d_labels.p + (size_t)i0 * n_anchors;
                                    cls_reduced |= 1u << l;
                                } else {
                                    at.t_box = d_boxes.p + (size_t)i0 * n_anchors * 4, at.t_w = ov[4], at.t_stride = ov[2];
                                    box_decoded |= 1u << l;
                                }
                            }
                            a0 += ov[3] * ov[4];
                        }
                    }
                    if (prof_conv) dev->prof_account(PROF_CONV, fl, by);
                    launch_conv_igemm(dtype, at, s);
                    ++oi;
                    continue;
                }
                fl = fl0, by = by0;
            }
            if (prof_conv) dev->prof_account(PROF_CONV, fl, by);
            launch_conv_igemm(dtype, a, s);
        } else {
            EltArgs a{};
            a.src = sb.p, a.dst = db.p, a.n = n, a.h = sb.h, a.w = sb.w, a.c = v[3];
            if (emb_host_out && oi + 1 == ops.size() && v[0] == OP_L2NORM && i0 == 0 && v[4] == outs[0].v[0]) a.dst = emb_host_out;
            a.s_cs = sb.c, a.s_coff = v[2], a.d_cs = db.c, a.d_coff = v[5];
            a.n_dev = n_items_dev;
            Prof pr(*dev, PROF_MISC, s, 0, 0);
            switch (v[0]) {
                case OP_SPPF_POOL: launch_sppf_pool(dtype, a, s); break;
                case OP_UPSAMPLE2X: launch_upsample2x(dtype, a, s); break;
                case OP_MAXPOOL3S2: launch_maxpool3s2(dtype, a, s); break;
                case OP_AVGPOOL: launch_avgpool(dtype, a, s); break;
                case OP_L2NORM: launch_l2norm(dtype, a, s); break;
                default: throw Error(AIC_ERR_FORMAT, "unknown op in engine file");
            }
        }
    }
    span_close();
}

DetArgs Model::det_args(int batch, float conf, float iou, int max_det, const LetterboxGeom* g) {
    AIC_REQUIRE(kind == KIND_YOLO, AIC_ERR_INVALID, "not a YOLO engine");
    AIC_REQUIRE(max_det > 0 && max_det <= max_det_cap, AIC_ERR_CAPACITY, "max_det out of range (1..1024)");
    AIC_REQUIRE(conf > 0.f && conf < 1.f, AIC_ERR_INVALID, "conf_thresh must be in (0,1)");
    DetArgs a{};
    int a0 = 0;
    for (size_t l = 0; l < outs.size(); ++l) {
        const int* v = outs[l].v;
        a.lvl[l].box = reinterpret_cast<const float*>(bufs[v[0]].p);
        a.lvl[l].cls = reinterpret_cast<const float*>(bufs[v[1]].p);
        a.lvl[l].stride = v[2], a.lvl[l].h = v[3], a.lvl[l].w = v[4], a.lvl[l].a0 = a0;
        a.lvl[l].cls_reduced = (cls_reduced >> l) & 1u, a.lvl[l].box_decoded = (box_decoded >> l) & 1u;
        a0 += v[3] * v[4];
    }
    a.n_levels = (int)outs.size(), a.n_anchors = n_anchors, a.nc = meta[0], a.reg_max = meta[1], a.batch = batch;
    a.fast_exp = dtype == AIC_F16;
    a.logit_thr = (float)std::log((double)conf / (1.0 - (double)conf));
    a.iou_thr = iou, a.max_det = max_det;
    a.pad_w = g ? g->pad_w : 0.f, a.pad_h = g ? g->pad_h : 0.f, a.ratio = g ? g->ratio : 1.f;
    a.orig_w = g ? g->src_w : in_w, a.orig_h = g ? g->src_h : in_h;
    a.boxes = d_boxes.p, a.max_logit = d_maxlogit.p, a.labels = d_labels.p, a.keys = nullptr;
    a.n_cand = d_ncand.p, a.num_dets = d_numdets.p;
    a.out_boxes = d_out_boxes.p, a.out_boxes_orig = g ? d_out_boxes_orig.p : nullptr;
    a.out_scores = d_out_scores.p, a.out_labels = d_out_labels.p;
    return a;
}

void Model::decode_nms(int batch, float conf, float iou, int max_det, const LetterboxGeom* g, hipStream_t s, char* host_out) {
    DetArgs a = det_args(batch, conf, iou, max_det, g);
    if (host_out) {
        AIC_REQUIRE(g != nullptr, AIC_ERR_INVALID, "decode_nms: host output wants the letterbox geometry");
        a.num_dets = reinterpret_cast<int*>(host_out);
        a.out_boxes_orig = reinterpret_cast<float*>(host_out + (size_t)batch * 4);
        a.out_scores = reinterpret_cast<float*>(host_out + (size_t)batch * 4 + (size_t)batch * max_det * 16);
        a.out_labels = reinterpret_cast<int*>(host_out + (size_t)batch * 4 + (size_t)batch * max_det * 20);
    }
    Prof pr(*dev, PROF_DET, s, 0, (double)batch * n_anchors * (4 * meta[1] + meta[0]) * 4);
    launch_decode(a, s);
    launch_select_sort_nms(a, s);
}

}  // namespace aic

// =================================================================================================
using namespace aic;

namespace {

// conf passed through Python is a double rounded to float; the logit threshold is computed from the
// fp64 value of that float so both sides of the parity test agree on the same number.
void copy_out(void* dst, const void* src, size_t bytes, int mem, hipStream_t s) {
    if (!bytes || !dst) return;
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, mem == AIC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
}

void load_input_nchw(Model& m, const float* images, int n, int mem, hipStream_t s) {
    const size_t cnt = (size_t)n * 3 * m.in_h * m.in_w;
    const float* src = images;
    if (mem == AIC_HOST) {
        m.d_in_f32.ensure(cnt);
        HIP_CHECK(hipMemcpyAsync(m.d_in_f32.p, images, cnt * 4, hipMemcpyHostToDevice, s));
        src = m.d_in_f32.p;
    }
    Prof pr(*m.dev, PROF_MISC, s);
    m.in_pix4 = false;
    launch_nchw_to_nhwc8(m.dtype, src, m.input(), n, m.in_h, m.in_w, s);
}

static void frame_sample(const uint8_t* f, size_t bytes, uint8_t* out) {
    const size_t step = (bytes - 64) / 63;
    for (int i = 0; i < 64; ++i) std::memcpy(out + 64 * i, f + (size_t)i * step, 64);
}

// `single`: one frame handed over in host memory -- the plugin loop's calls; see Device::FrameCache
const uint8_t* stage_frames(Model& m, const uint8_t* frames, size_t bytes, int mem, hipStream_t s, bool single = false) {
    if (mem == AIC_DEVICE) return frames;
    static const bool cache_on = getenv("AICAM_NO_FRAME_CACHE") == nullptr;
    Device::FrameCache& fc = m.dev->frame_cache;
    uint8_t smp[64 * 64];
    const bool cacheable = cache_on && single && bytes >= 64 * 64 && s == m.dev->s_main;
    if (cacheable) {
        frame_sample(frames, bytes, smp);
        if (fc.dev && fc.owner != &m && fc.bytes == bytes && std::memcmp(smp, fc.sample, sizeof smp) == 0) {
            m.dev->frame_cache_hits += 1;
            return fc.dev;                   // uploaded by the other engine's call a moment ago
        }
    }
    m.d_frames.ensure(bytes + 16);           // (slack: the fused crop reads aligned 12-byte groups that may end past the last pixel)
    HIP_CHECK(hipMemcpyAsync(m.d_frames.p, frames, bytes, hipMemcpyHostToDevice, s));
    if (cacheable) {
        fc.dev = m.d_frames.p, fc.owner = &m, fc.bytes = bytes;
        std::memcpy(fc.sample, smp, sizeof smp);
    } else if (fc.owner == &m) {
        fc.dev = nullptr;                    // this engine's staging buffer now holds something else
    }
    return m.d_frames.p;
}

}  // namespace

extern "C" {

int aic_model_load_mem(const void* blob, size_t nbytes, int device_id, int dtype, int max_items, aic_model** out) {
    return guarded([&] {
        AIC_REQUIRE(blob && out, AIC_ERR_INVALID, "NULL argument");
        *out = new aic_model(device(device_id), blob, nbytes, dtype, max_items);
    });
}

int aic_model_load(const char* path, int device_id, int dtype, int max_items, aic_model** out) {
    return guarded([&] {
        AIC_REQUIRE(path && out, AIC_ERR_INVALID, "NULL argument");
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        AIC_REQUIRE(f.good(), AIC_ERR_NOT_FOUND, std::string("engine file not found: ") + path);
        const size_t n = (size_t)f.tellg();
        std::vector<char> blob(n);
        f.seekg(0);
        f.read(blob.data(), (std::streamsize)n);
        AIC_REQUIRE(f.good(), AIC_ERR_FORMAT, std::string("cannot read engine file: ") + path);
        *out = new aic_model(device(device_id), blob.data(), n, dtype, max_items);
    });
}

int aic_model_read_buffer(aic_model* m, int buf, void* out, size_t bytes) {
    return guarded([&] {
        AIC_REQUIRE(m && out && buf >= 0 && buf < (int)m->m.bufs.size() && m->m.bufs[buf].p, AIC_ERR_INVALID, "bad buffer index");
        AIC_REQUIRE(bytes <= m->m.bufs[buf].per_item * (size_t)m->m.max_items, AIC_ERR_CAPACITY, "more bytes than the buffer holds");
        m->m.dev->use();
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out, m->m.bufs[buf].p, bytes, hipMemcpyDeviceToHost));
    });
}

int aic_model_destroy(aic_model* m) {
    return guarded([&] {
        if (m) {
            m->m.dev->use();
            (void)hipDeviceSynchronize();
            if (m->m.dev->frame_cache.owner == &m->m) m->m.dev->frame_cache = Device::FrameCache{};     // its staging buffer goes away
        }
        delete m;
    });
}

int aic_model_info(const aic_model* m, int* kind, int* in_h, int* in_w, int* out_dim, int* n_anchors,
                   double* flops_per_item, int* n_convs) {
    return guarded([&] {
        AIC_REQUIRE(m, AIC_ERR_INVALID, "NULL model");
        if (kind) *kind = m->m.kind;
        if (in_h) *in_h = m->m.in_h;
        if (in_w) *in_w = m->m.in_w;
        if (out_dim) *out_dim = m->m.out_dim;
        if (n_anchors) *n_anchors = m->m.n_anchors;
        if (flops_per_item) *flops_per_item = m->m.flops_per_item;
        if (n_convs) *n_convs = m->m.n_convs;
    });
}

int aic_yolo_infer(aic_model* mm, const float* images, int batch, int mem, float conf, float iou, int max_det,
                   int32_t* num_dets, float* bboxes, float* scores, int32_t* labels) {
    return guarded([&] {
        AIC_REQUIRE(mm && images && batch > 0, AIC_ERR_INVALID, "bad argument");
        Model& m = mm->m;
        AIC_REQUIRE(m.kind == KIND_YOLO, AIC_ERR_INVALID, "not a YOLO engine");
        m.dev->use();
        hipStream_t s = m.dev->s_main;
        load_input_nchw(m, images, batch, mem, s);
        m.reduce_cls = true;                    // (only the decoded outputs leave this call)
        m.side_ok = true;
        m.run(batch, s);
        m.side_ok = false;
        m.decode_nms(batch, conf, iou, max_det, nullptr, s);
        copy_out(num_dets, m.d_numdets.p, (size_t)batch * 4, mem, s);
        copy_out(bboxes, m.d_out_boxes.p, (size_t)batch * max_det * 16, mem, s);
        copy_out(scores, m.d_out_scores.p, (size_t)batch * max_det * 4, mem, s);
        copy_out(labels, m.d_out_labels.p, (size_t)batch * max_det * 4, mem, s);
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_yolo_head(aic_model* mm, const float* images, int batch, int mem, float* dfl, float* cls) {
    return guarded([&] {
        AIC_REQUIRE(mm && images && batch > 0, AIC_ERR_INVALID, "bad argument");
        Model& m = mm->m;
        AIC_REQUIRE(m.kind == KIND_YOLO, AIC_ERR_INVALID, "not a YOLO engine");
        m.dev->use();
        hipStream_t s = m.dev->s_main;
        load_input_nchw(m, images, batch, mem, s);
        m.reduce_cls = false;                   // the raw class logits are what this call returns
        m.side_ok = true;
        m.run(batch, s);
        m.side_ok = false;
        const int nc = m.meta[0], nb = 4 * m.meta[1];
        int a0 = 0;
        for (auto& o : m.outs) {
            const int hw = o.v[3] * o.v[4];
            const float* bsrc = reinterpret_cast<const float*>(m.bufs[o.v[0]].p);
            const float* csrc = reinterpret_cast<const float*>(m.bufs[o.v[1]].p);
            for (int b = 0; b < batch; ++b) {
                if (dfl) HIP_CHECK(hipMemcpyAsync(dfl + ((size_t)b * m.n_anchors + a0) * nb, bsrc + (size_t)b * hw * nb,
                                                  (size_t)hw * nb * 4, hipMemcpyDeviceToHost, s));
                if (cls) HIP_CHECK(hipMemcpyAsync(cls + ((size_t)b * m.n_anchors + a0) * nc, csrc + (size_t)b * hw * nc,
                                                  (size_t)hw * nc * 4, hipMemcpyDeviceToHost, s));
            }
            a0 += hw;
        }
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_yolo_decode(aic_model* mm, const float* images, int batch, int mem, float* boxes, float* max_logit, int32_t* labels) {
    return guarded([&] {
        AIC_REQUIRE(mm && images && batch > 0, AIC_ERR_INVALID, "bad argument");
        Model& m = mm->m;
        AIC_REQUIRE(m.kind == KIND_YOLO, AIC_ERR_INVALID, "not a YOLO engine");
        m.dev->use();
        hipStream_t s = m.dev->s_main;
        load_input_nchw(m, images, batch, mem, s);
        m.reduce_cls = true;
        m.side_ok = true;
        m.run(batch, s);
        m.side_ok = false;
        const DetArgs a = m.det_args(batch, 0.5f, 0.5f, 1, nullptr);
        launch_decode(a, s);
        const size_t ba = (size_t)batch * m.n_anchors;
        copy_out(boxes, m.d_boxes.p, ba * 16, AIC_HOST, s);
        copy_out(max_logit, m.d_maxlogit.p, ba * 4, AIC_HOST, s);
        copy_out(labels, m.d_labels.p, ba * 4, AIC_HOST, s);
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_reid_infer(aic_model* mm, const float* crops, int n, int mem, float* emb, int out_mem) {
    return guarded([&] {
        AIC_REQUIRE(mm && n >= 0, AIC_ERR_INVALID, "bad argument");
        if (n == 0) return;
        AIC_REQUIRE(crops && emb, AIC_ERR_INVALID, "NULL argument");
        Model& m = mm->m;
        AIC_REQUIRE(m.kind == KIND_REID, AIC_ERR_INVALID, "not a ReID engine");
        m.dev->use();
        hipStream_t s = m.dev->s_main;
        // more crops than the activation arena holds: launch groups of max_items (reid_model.py:80-101 passes every valid
        // crop; a crowded frame must not fail or drop detections)
        const size_t per_in = (size_t)3 * m.in_h * m.in_w;
        for (int c0 = 0; c0 < n; c0 += m.max_items) {
            const int k = std::min(m.max_items, n - c0);
            load_input_nchw(m, crops + (size_t)c0 * per_in, k, mem, s);
            m.run(k, s);
            copy_out(emb + (size_t)c0 * m.out_dim, m.embeddings(), (size_t)k * m.out_dim * 4, out_mem, s);
        }
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_letterbox(int device_id, const uint8_t* frame, int h, int w, int out_h, int out_w, float* out, float* ratio,
                  float* pad_w, float* pad_h) {
    return guarded([&] {
        AIC_REQUIRE(frame && out && h > 0 && w > 0 && out_h > 0 && out_w > 0, AIC_ERR_INVALID, "bad argument");
        Device& d = device(device_id);
        hipStream_t s = d.s_main;
        const LetterboxGeom g = letterbox_geometry(h, w, out_h, out_w);
        DevBuf<uint8_t> df((size_t)h * w * 3);
        DevBuf<float> dout((size_t)3 * out_h * out_w);
        HIP_CHECK(hipMemcpyAsync(df.p, frame, df.n, hipMemcpyHostToDevice, s));
        {
            Prof pr(d, PROF_LETTERBOX, s, 0, (double)h * w * 3 + 12.0 * out_h * out_w);
            launch_letterbox(df.p, 1, g, 0, AIC_F32, dout.p, s);
        }
        HIP_CHECK(hipMemcpyAsync(out, dout.p, dout.n * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (ratio) *ratio = g.ratio;
        if (pad_w) *pad_w = g.pad_w;
        if (pad_h) *pad_h = g.pad_h;
    });
}

int aic_letterbox_image(int device_id, const uint8_t* frame, int h, int w, int unpad_h, int unpad_w, int top, int bottom, int left, int right,
                        int color_b, int color_g, int color_r, uint8_t* out) {
    return guarded([&] {
        AIC_REQUIRE(frame && out && h > 0 && w > 0 && unpad_h > 0 && unpad_w > 0, AIC_ERR_INVALID, "bad argument");
        AIC_REQUIRE(top >= 0 && bottom >= 0 && left >= 0 && right >= 0, AIC_ERR_INVALID, "negative border");   // cv2.copyMakeBorder refuses them too
        Device& d = device(device_id);
        hipStream_t s = d.s_main;
        LetterboxGeom g{};
        g.src_h = h, g.src_w = w, g.unpad_h = unpad_h, g.unpad_w = unpad_w, g.top = top, g.left = left;
        g.out_h = unpad_h + top + bottom, g.out_w = unpad_w + left + right;
        auto u8 = [](int v) { return v < 0 ? 0 : v > 255 ? 255 : v; };
        const int color[3] = {u8(color_b), u8(color_g), u8(color_r)};
        DevBuf<uint8_t> df((size_t)h * w * 3), dout((size_t)g.out_h * g.out_w * 3);
        HIP_CHECK(hipMemcpyAsync(df.p, frame, df.n, hipMemcpyHostToDevice, s));
        launch_letterbox_u8(df.p, g, color, dout.p, s);
        HIP_CHECK(hipMemcpyAsync(out, dout.p, dout.n, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_crop_resize(int device_id, const uint8_t* frame, int h, int w, const float* boxes, int n, int out_h, int out_w,
                    float* out, int32_t* valid) {
    return guarded([&] {
        AIC_REQUIRE(n >= 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, AIC_ERR_INVALID, "bad argument");
        if (n == 0) return;
        AIC_REQUIRE(frame && boxes && out, AIC_ERR_INVALID, "NULL argument");
        Device& d = device(device_id);
        hipStream_t s = d.s_main;
        DevBuf<uint8_t> df((size_t)h * w * 3);
        DevBuf<float> db((size_t)n * 4), dout((size_t)n * 3 * out_h * out_w);
        DevBuf<int> dv(n);
        HIP_CHECK(hipMemcpyAsync(df.p, frame, df.n, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(db.p, boxes, (size_t)n * 16, hipMemcpyHostToDevice, s));
        {
            Prof pr(d, PROF_CROP, s, 0, (double)n * out_h * out_w * 15);
            launch_crop_resize(df.p, h, w, db.p, nullptr, n, nullptr, out_h, out_w, 0, AIC_F32, dout.p, dv.p, s);
        }
        HIP_CHECK(hipMemcpyAsync(out, dout.p, dout.n * 4, hipMemcpyDeviceToHost, s));
        if (valid) HIP_CHECK(hipMemcpyAsync(valid, dv.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_detect(aic_model* mm, const uint8_t* frames, int batch, int h, int w, int mem, float conf, float iou, int max_det,
               int32_t* num_dets, float* boxes, float* scores, int32_t* labels) {
    return guarded([&] {
        AIC_REQUIRE(mm && frames && batch > 0 && h > 0 && w > 0, AIC_ERR_INVALID, "bad argument");
        Model& m = mm->m;
        AIC_REQUIRE(m.kind == KIND_YOLO, AIC_ERR_INVALID, "not a YOLO engine");
        AIC_REQUIRE(batch <= m.max_items, AIC_ERR_CAPACITY, "batch exceeds the engine's max_items");
        m.dev->use();
        hipStream_t s = m.dev->s_main;
        const uint8_t* df = stage_frames(m, frames, (size_t)batch * h * w * 3, mem, s, batch == 1);
        const LetterboxGeom g = letterbox_geometry(h, w, m.in_h, m.in_w);
        m.reduce_cls = true;
        m.side_ok = true;
        m.run_frames(df, batch, g, s);
        m.side_ok = false;
        static const bool direct = getenv("AICAM_NO_DET_HOST_OUT") == nullptr;
        if (direct && batch <= 4 && num_dets && boxes && scores && labels) {
            // the per-frame plugin loop: the NMS kernel stores what it keeps into page-locked host memory itself
            const size_t per = (size_t)max_det * 24;
            m.h_det.ensure((size_t)batch * (4 + per) + 16);
            m.decode_nms(batch, conf, iou, max_det, &g, s, m.h_det.p);
            HIP_CHECK(hipStreamSynchronize(s));
            const int32_t* hn = reinterpret_cast<const int32_t*>(m.h_det.p);
            const char* hb = m.h_det.p + (size_t)batch * 4;
            const char* hs = hb + (size_t)batch * max_det * 16;
            const char* hl = hb + (size_t)batch * max_det * 20;
            for (int b = 0; b < batch; ++b) {          // rows past a frame's count are left as the caller passed them
                const int n = std::min(std::max(hn[b], 0), max_det);
                num_dets[b] = hn[b];
                std::memcpy(boxes + (size_t)b * max_det * 4, hb + (size_t)b * max_det * 16, (size_t)n * 16);
                std::memcpy(scores + (size_t)b * max_det, hs + (size_t)b * max_det * 4, (size_t)n * 4);
                std::memcpy(labels + (size_t)b * max_det, hl + (size_t)b * max_det * 4, (size_t)n * 4);
            }
            return;
        }
        m.decode_nms(batch, conf, iou, max_det, &g, s);
        copy_out(num_dets, m.d_numdets.p, (size_t)batch * 4, AIC_HOST, s);
        copy_out(boxes, m.d_out_boxes_orig.p, (size_t)batch * max_det * 16, AIC_HOST, s);
        copy_out(scores, m.d_out_scores.p, (size_t)batch * max_det * 4, AIC_HOST, s);
        copy_out(labels, m.d_out_labels.p, (size_t)batch * max_det * 4, AIC_HOST, s);
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int aic_reid_embed(aic_model* mm, const uint8_t* frame, int h, int w, int mem, const float* boxes, int n, float* emb,
                   int32_t* valid) {
    return guarded([&] {
        AIC_REQUIRE(mm && n >= 0 && h > 0 && w > 0, AIC_ERR_INVALID, "bad argument");
        if (n == 0) return;
        AIC_REQUIRE(frame && boxes && emb, AIC_ERR_INVALID, "NULL argument");
        Model& m = mm->m;
        AIC_REQUIRE(m.kind == KIND_REID, AIC_ERR_INVALID, "not a ReID engine");
        m.dev->use();
        hipStream_t s = m.dev->s_main;
        const uint8_t* df = stage_frames(m, frame, (size_t)h * w * 3, mem, s, true);
        m.d_crop_boxes.ensure((size_t)n * 4);
        m.d_valid.ensure(n);
        HIP_CHECK(hipMemcpyAsync(m.d_crop_boxes.p, boxes, (size_t)n * 16, hipMemcpyHostToDevice, s));
        m.in_pix4 = m.input_pix4_ok();
        static const bool fuse_crop = getenv("AICAM_NO_FUSE_CROP") == nullptr;
        static const bool direct = getenv("AICAM_NO_EMB_HOST_OUT") == nullptr;
        const BufDesc& eb = m.bufs[m.outs[0].v[0]];
        if (direct && fuse_crop && n <= m.max_items && n <= 512 && m.in_pix4 && m.in_h <= 192 && mem != AIC_DEVICE && eb.f32 &&
            eb.c == m.out_dim && m.outs[0].v[2] == 0 && m.ops.back().v[0] == OP_L2NORM && m.ops.back().v[5] == 0) {
            // the per-frame plugin loop: embeddings and crop validity are stored into page-locked host memory by the kernels that produce them
            const size_t eb_bytes = (size_t)n * m.out_dim * 4;
            m.h_emb.ensure(eb_bytes + (size_t)n * 4 + 16);
            int* hv = reinterpret_cast<int*>(m.h_emb.p + eb_bytes);
            m.crop_src = CropSrc{df, h, w, m.d_crop_boxes.p, nullptr, hv};
            m.emb_host_out = reinterpret_cast<float*>(m.h_emb.p);
            try {
                m.run(n, s);
            } catch (...) {
                m.crop_src.frames = nullptr, m.emb_host_out = nullptr;
                throw;
            }
            m.crop_src.frames = nullptr, m.emb_host_out = nullptr;
            HIP_CHECK(hipStreamSynchronize(s));
            std::memcpy(emb, m.h_emb.p, eb_bytes);
            if (valid) std::memcpy(valid, hv, (size_t)n * 4);
            return;
        }
        for (int c0 = 0; c0 < n; c0 += m.max_items) {   // launch groups of max_items: every detection is embedded (deepsort_tracker.py:104-113)
            const int k = std::min(m.max_items, n - c0);
            if (fuse_crop && m.in_pix4 && m.in_h <= 192 && mem != AIC_DEVICE) {
                // crop + resize + normalise inside the stem kernel, as the pipeline does it (same arithmetic, pixel for pixel): one launch fewer
                // per call of the per-frame plugin loop, and no crop tensor.  (A caller's own device buffer keeps the separate crop kernel:
                // the fused form's aligned 12-byte reads want 16 bytes of slack behind the frame, which only our staging buffer promises.)
                m.crop_src = CropSrc{df, h, w, m.d_crop_boxes.p + (size_t)c0 * 4, nullptr, m.d_valid.p + c0};
            } else {
                Prof pr(*m.dev, PROF_CROP, s, 0, (double)k * m.in_h * m.in_w * 19);
                launch_crop_resize(df, h, w, m.d_crop_boxes.p + (size_t)c0 * 4, nullptr, k, nullptr, m.in_h, m.in_w, m.in_pix4 ? 2 : 1, m.dtype,
                                   m.input(), m.d_valid.p + c0, s);
            }
            m.run(k, s);
            m.crop_src.frames = nullptr;
            copy_out(emb + (size_t)c0 * m.out_dim, m.embeddings(), (size_t)k * m.out_dim * 4, AIC_HOST, s);
        }
        if (valid) HIP_CHECK(hipMemcpyAsync(valid, m.d_valid.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

}  // extern "C"
