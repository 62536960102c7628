// READ's MIMO-UNet-style refinement CNN (READ/models/unet.py:121-285) as a pre-built launch plan.
//
// The network is fixed by the reference (get_net(): UNet(8, 3, feature_scale=4, num_res=4),
// READ/pipelines/ogl.py:19-27; base_channel 32, unet.py:141), so the plan is built once per
// resolution: 99 fused gated-conv launches (+ 3 for the AFF inputs of coarser levels) + 3 bilinear x4 upsamples, every torch.cat /
// F.interpolate(nearest) / FAM multiply / residual add folded into a conv's loader or epilogue.
// Activations are NHWC fp32 in a caller-provided workspace; one C-ABI call enqueues a frame.
#include <math.h>

#include <string>
#include <vector>

#include "common.h"

namespace readhip {
int launch_gated_conv(const read_conv_desc *d, hipStream_t stream);
int conv_uses_wino(const read_conv_desc *d);
int conv_uses_w4(const read_conv_desc *d);
int conv_uses_w4h(const read_conv_desc *d);
int conv_uses_d3h(const read_conv_desc *d);
int conv_uses_d3h_s2(const read_conv_desc *d);
}
using namespace readhip;

namespace {

constexpr int BASE = 32;      // base_channel, unet.py:141
constexpr int NUM_RES = 4;    // num_res, ogl.py:24
constexpr int IN_CH = READ_DESC_CHANNELS;

struct LayerInfo {
    std::string path;
    int cin, cout, k, stride, elu, kc;
    size_t raw_off, w_off, p_off;   // float offsets into the raw / packed blobs
    size_t wino_off;                // Winograd-transformed weights of 3x3/s1 layers with Cin % 16 == 0, else NO_WINO
    size_t w16_off;                 // ... in the order of the wave-autonomous Winograd kernel (read_conv_pack_w16_host)
    size_t w4_off;                  // Winograd F(4x4,3x3) weights of the 3x3/s1 layers with Cin >= 32 and Cout % 32 == 0, else NO_WINO
    size_t sc_off = ~(size_t)0;     // small-Cout order of the 3x3/s1 layers with Cout <= 4 (the output layer), else NO_WINO
    size_t w4h_off = ~(size_t)0;    // F(4x4) weights split into f16 piece pairs (read_conv_pack_w4h_host) of the w4 layers with Cin % 32 == 0, else NO_WINO
    size_t d3h_off = ~(size_t)0;    // the plain 3x3 weights as f16 piece pairs (read_conv_pack_d3h_host) of the same layers, else NO_WINO
    size_t t3h_off = ~(size_t)0;    // 3x3 / stride-1 layers over the 8-channel pyramid: the implicit-GEMM operand (read_conv_pack_t3h_host), else NO_WINO
};
constexpr size_t NO_WINO = ~(size_t)0;

// A 1x1 layer whose weights are a block of existing layers' weights: input channels [ci0, ci0 + cin) of the parts'
// concatenated input, the parts' output channels stacked.  Used to evaluate the coarse-level inputs of the AFF 1x1
// convs at their own resolution (a 1x1 conv commutes with the nearest up-sampling of unet.py:239-254).
struct DerivedInfo {
    std::string name;
    int cin, cout, ci0;
    std::vector<int> parts;         // indices into Arch::layers
    size_t w_off, p_off;            // packed weights; parameters (zero biases) — gated finals use their part's own p_off
    size_t h_off = ~(size_t)0;      // the same weights as f16 piece pairs (read_conv_pack_dkh_host, ksize 1: the split-operand pixel-lane kernel)
};

struct Arch {
    std::vector<LayerInfo> layers;
    std::vector<DerivedInfo> derived;
    size_t raw_floats = 0, packed_floats = 0;
    int find_derived(const std::string &p) const
    {
        for (size_t i = 0; i < derived.size(); ++i)
            if (derived[i].name == p) return (int)i;
        return -1;
    }
    int find(const std::string &p) const
    {
        for (size_t i = 0; i < layers.size(); ++i)
            if (layers[i].path == p) return (int)i;
        return -1;
    }
};

size_t raw_layer_floats(int cin, int cout, int k) { return 2 * ((size_t)cout * cin * k * k + cout) + 4 * (size_t)cout; }

// Layout of the packed blob.  FULL: every fragment order of every layer (direct + both F(2x2) orders + F(4x4) where they exist):
// any tuning knob can send a layer to any of its kernels — 952 MB for a 121 MB model.  LEAN: what the default plan reads —
// a layer the F(4x4) kernel takes carries its F(4x4) order ONLY (that kernel runs 73 of the 105 launches; its layers' other three
// orders were 500 MB nobody touched), the layers no launch executes (ConvsOut) carry nothing; everything else as in FULL.
// The raw blob (read_unet_raw_floats) is the same for both.
Arch build_arch(int layout)
{
        const bool lean = layout == READ_UNET_LAYOUT_LEAN;
        Arch a;
        auto add = [&](const std::string &path, int cin, int cout, int k, int s, int elu, int kc) {
            LayerInfo L{path, cin, cout, k, s, elu, kc, 0, NO_WINO, 0, NO_WINO, NO_WINO, NO_WINO};
            L.raw_off = a.raw_floats;
            a.raw_floats += raw_layer_floats(cin, cout, k);
            const bool wino = k == 3 && s == 1 && kc == 16 && cin % 16 == 0, w4 = wino && cin >= 32 && cout % 32 == 0;
            const bool unused = path.compare(0, 9, "ConvsOut.") == 0;
            const bool d3h_s2 = (k == 3 || k == 4) && s == 2 && cin % 32 == 0 && cout % 32 == 0;   // down-sampling (3x3) and decoder (4x4) stride-2 layers: direct split-operand kernel
            L.p_off = a.packed_floats;
            a.packed_floats += read_conv_param_floats(cout);
            if (!(lean && (w4 || unused || d3h_s2))) {
                L.w_off = a.packed_floats;
                a.packed_floats += read_conv_packed_floats(cin, cout, k);
            }
            if (wino && !(lean && (w4 || unused))) {
                L.wino_off = a.packed_floats;
                a.packed_floats += read_conv_wino_floats(cin, cout);
                L.w16_off = a.packed_floats;
                a.packed_floats += read_conv_wino_floats(cin, cout);
            }
            // the Winograd split-operand kernel (f16 matrix cores) runs the layer by default — except FAM's x1 * x2 launches, which the
            // DIRECT split-operand kernel takes
            const bool fam = path.compare(0, 3, "FAM") == 0;
            const bool w4h = w4 && cin % 32 == 0 && !fam, d3h = (w4 && cin % 32 == 0) || d3h_s2;
            if (w4 && !(lean && (unused || w4h || (d3h && fam)))) {
                L.w4_off = a.packed_floats;
                a.packed_floats += read_conv_w4_floats(cin, cout);
            }
            if (w4h && !(lean && unused)) {
                L.w4h_off = a.packed_floats;
                a.packed_floats += read_conv_w4h_floats(cin, cout);
            }
            if (d3h && !(lean && (unused || !(fam || d3h_s2)))) {
                L.d3h_off = a.packed_floats;
                a.packed_floats += read_conv_dkh_floats(cin, cout, k);
            }
            // 1x1 layers: the operand of the split-operand pixel-lane kernel (both layouts: the fp32 order stays beside it as the fallback)
            if (k == 1 && s == 1 && cin <= 256 && read_conv_dkh_floats(cin, cout, 1)) {
                L.d3h_off = a.packed_floats;
                a.packed_floats += read_conv_dkh_floats(cin, cout, 1);
            }
            if (k == 3 && s == 1 && cin == IN_CH && read_conv_t3h_floats(cin, cout)) {
                L.t3h_off = a.packed_floats;
                a.packed_floats += read_conv_t3h_floats(cin, cout);
            }
            if (k == 3 && s == 1 && read_conv_sc_floats(cin, cout) && !(lean && unused)) {
                a.packed_floats = (a.packed_floats + 15) / 16 * 16;     // 64-byte aligned: scalar loads of 16 dwords
                L.sc_off = a.packed_floats;
                a.packed_floats += read_conv_sc_floats(cin, cout);
            }
            a.layers.push_back(L);
        };
        // SCM (unet.py:92-106): SCM2 -> 64 planes @1/2, SCM1 -> 128 @1/4, SCM0 -> 256 @1/8
        const int scm_planes[3] = {BASE * 8, BASE * 4, BASE * 2};   // SCM0, SCM1, SCM2
        for (int n = 0; n < 3; ++n) {
            const int P = scm_planes[n];
            const std::string p = "SCM" + std::to_string(n);
            add(p + ".main.0", IN_CH, P / 4, 3, 1, 1, 8);
            add(p + ".main.1", P / 4, P / 2, 1, 1, 1, 16);
            add(p + ".main.2", P / 2, P / 2, 3, 1, 1, 16);
            add(p + ".main.3", P / 2, P - IN_CH, 1, 1, 1, 16);
            add(p + ".conv", P, P, 1, 1, 0, 8);                     // cat[x(8), main(P-8)]
        }
        // feat_extract (unet.py:156-165)
        add("feat_extract.0", IN_CH, BASE, 3, 1, 1, 8);
        add("feat_extract.1", BASE, BASE * 2, 3, 2, 1, 16);
        add("feat_extract.2", BASE * 2, BASE * 4, 3, 2, 1, 16);
        add("feat_extract.3", BASE * 4, BASE * 2, 4, 2, 1, 16);
        add("feat_extract.4", BASE * 2, BASE, 4, 2, 1, 16);
        add("feat_extract.5", BASE, 3, 3, 1, 0, 16);
        add("feat_extract.6", BASE * 4, BASE * 8, 3, 2, 1, 16);
        add("feat_extract.7", BASE * 8, BASE * 4, 4, 2, 1, 16);
        // Encoder / Decoder: 4 blocks x NUM_RES ResBlocks x 2 BasicConvs (unet.py:11-20,56-76)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < NUM_RES; ++j) {
                const int C = BASE << i;
                const std::string p = "Encoder." + std::to_string(i) + ".layers." + std::to_string(j) + ".main.";
                add(p + "0", C, C, 3, 1, 1, 16);
                add(p + "1", C, C, 3, 1, 0, 16);
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < NUM_RES; ++j) {
                const int C = (BASE * 8) >> i;
                const std::string p = "Decoder." + std::to_string(i) + ".layers." + std::to_string(j) + ".main.";
                add(p + "0", C, C, 3, 1, 1, 16);
                add(p + "1", C, C, 3, 1, 0, 16);
            }
        add("Convs.0", BASE * 8, BASE * 4, 1, 1, 1, 16);
        add("Convs.1", BASE * 4, BASE * 2, 1, 1, 1, 16);
        add("Convs.2", BASE * 2, BASE, 1, 1, 1, 16);
        add("ConvsOut.0", BASE * 4, 3, 3, 1, 0, 16);   // never executed (unet.py:181-186)
        add("ConvsOut.1", BASE * 2, 3, 3, 1, 0, 16);
        for (int i = 0; i < 3; ++i) {
            const std::string p = "AFFs." + std::to_string(i) + ".conv.";
            add(p + "0", BASE * 15, BASE << i, 1, 1, 1, 16);
            add(p + "1", BASE << i, BASE << i, 3, 1, 0, 16);
        }
        add("FAM0.merge", BASE * 8, BASE * 8, 3, 1, 0, 16);
        add("FAM1.merge", BASE * 4, BASE * 4, 3, 1, 0, 16);
        add("FAM2.merge", BASE * 2, BASE * 2, 3, 1, 0, 16);
        // AFF first convs split by the level their inputs live at; concat order res1(32) res2(64) res3(128) z(256)
        auto derive = [&](const std::string &name, int ci0, int cin, std::vector<int> affs) {
            DerivedInfo D{name, cin, 0, ci0, {}, 0, 0};
            for (int k : affs) {
                const int li = a.find("AFFs." + std::to_string(k) + ".conv.0");
                D.parts.push_back(li);
                D.cout += a.layers[li].cout;
            }
            D.w_off = a.packed_floats;
            a.packed_floats += read_conv_packed_floats(cin, D.cout, 1);
            D.p_off = a.packed_floats;
            a.packed_floats += read_conv_param_floats(D.cout);
            if (cin <= 256 && read_conv_dkh_floats(cin, D.cout, 1)) {
                D.h_off = a.packed_floats;
                a.packed_floats += read_conv_dkh_floats(cin, D.cout, 1);
            }
            a.derived.push_back(D);
        };
        derive("AFFq3", BASE * 7, BASE * 8, {0, 1, 2});   // z    @1/8 -> partial sums of AFF0, AFF1, AFF2
        derive("AFFq2", BASE * 3, BASE * 4, {0, 1});      // res3 @1/4 -> AFF0, AFF1
        derive("AFFq1", BASE, BASE * 2, {0});             // res2 @1/2 -> AFF0
        derive("AFFs.0.conv.0r", 0, BASE, {0});           // what is left at the layer's own level
        derive("AFFs.1.conv.0r", 0, BASE * 3, {1});
        derive("AFFs.2.conv.0r", 0, BASE * 7, {2});
        // Convs.k = 1x1 over cat[Upsample4_bilinear(fe), r] (unet.py:261-262,269-270,277-278): the half that multiplies the up-sampled
        // tensor is applied at ITS level (1/16 of the pixels, `linear`) and enters the gated launch as a bilinear pre-activation addend
        auto derive_of = [&](const std::string &name, const std::string &layer, int ci0, int cin) {
            const int li = a.find(layer);
            DerivedInfo D{name, cin, a.layers[li].cout, ci0, {li}, 0, 0};
            D.w_off = a.packed_floats;
            a.packed_floats += read_conv_packed_floats(cin, D.cout, 1);
            D.p_off = a.packed_floats;
            a.packed_floats += read_conv_param_floats(D.cout);
            if (cin <= 256 && read_conv_dkh_floats(cin, D.cout, 1)) {
                D.h_off = a.packed_floats;
                a.packed_floats += read_conv_dkh_floats(cin, D.cout, 1);
            }
            a.derived.push_back(D);
        };
        for (int k = 0; k < 3; ++k) {
            const int Cu = (BASE * 4) >> k;                                  // channels of up4(fe) = channels of r
            const std::string L = "Convs." + std::to_string(k);
            derive_of(L + ".u", L, 0, Cu);
            derive_of(L + ".r", L, Cu, Cu);
        }
        return a;
}

const Arch &arch(int layout = READ_UNET_LAYOUT_FULL)
{
    static const Arch A[2] = {build_arch(READ_UNET_LAYOUT_FULL), build_arch(READ_UNET_LAYOUT_LEAN)};
    return A[layout == READ_UNET_LAYOUT_LEAN ? 1 : 0];
}

// ------------------------------------------------------------------------------------------
struct Tensor {
    std::string name;
    float *p = nullptr;   // null for external tensors until forward()
    int ext = -1;         // 0..3 = x0..x3, 4 = rgb output
    int H = 0, W = 0, C = 0;
    int lane = 0;         // lane of the op that writes it
};

struct Op {
    enum Kind { CONV, UP4 } kind;
    read_conv_desc d;          // CONV
    int src_t[READ_CONV_MAX_SRC], mul_t, res_t, out_t;   // tensor ids (for external patching)
    int in_t;                  // UP4
    int pre_t = -1;            // CONV: tensor of the pre-activation addend
    double flops;
    int is_c3s1;
    int lane = 0;              // 0: the caller's stream; 1..3: side stream of SCM 0..2 (independent of the trunk)
    int wait_mask = 0;         // side lanes whose results this (main-lane) op consumes
    std::string label;
};

}  // namespace

struct read_unet {
    int H, W;
    int layout = READ_UNET_LAYOUT_FULL;
    const float *packed;
    char *ws;
    size_t ws_bytes, ws_used;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    hipEvent_t *events = nullptr;
    int n_events = 0;
    // the three SCM chains (15 small launches) run on side streams, concurrently with each other and with the first
    // trunk layers; joined by events right before FAM2 / FAM1 / FAM0 read them
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_start = nullptr, ev_done[3] = {nullptr, nullptr, nullptr};
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int g_unet_aff_split = 1;     // read_tuning_set("unet_aff_split", 0): the AFF first convs as single 480-channel launches
// read_tuning_set("unet_up_fold", 1): the bilinear x4 up-sampling folded into the Convs.k launches (read_conv_desc.pre_bilinear).
// Built and measured in round 5 (profiles/r5_up_fold_ab.md): results equal (148 dB either way) and 110 MB per frame less written and
// re-read, but NOT faster — the three tiny `Convs.k.u` launches cost 14 - 22 us each (launch floor of the 1x1 kernels) and the
// bilinear epilogue gathers 32 float4 per lane: Convs.0 / 1 / 2 take 64.8 / 63.6 / 102.0 us folded against 61.6 / 70.2 / 83.9 with
// the separate pass (non-family launches 1.346 vs 1.332 ms per frame, 220.8 vs 220.4 frames/s).  Default off.
int g_unet_up_fold = 0;

struct Builder {
    read_unet *u;
    bool dry;                   // size-only pass
    size_t used = 0;
    int cur_lane = 0;

    int tensor(const std::string &name, int level, int C)
    {
        Tensor t;
        t.name = name;
        t.H = u->H >> level;
        t.W = u->W >> level;
        t.C = C;
        const size_t bytes = align_up((size_t)t.H * t.W * C * sizeof(float), 256);
        if (!dry) t.p = reinterpret_cast<float *>(u->ws + used);
        used += bytes;
        u->tensors.push_back(t);
        return (int)u->tensors.size() - 1;
    }
    int external(const std::string &name, int ext, int level, int C)
    {
        Tensor t;
        t.name = name;
        t.ext = ext;
        t.H = u->H >> level;
        t.W = u->W >> level;
        t.C = C;
        u->tensors.push_back(t);
        return (int)u->tensors.size() - 1;
    }

    struct PreRef {
        int t = -1, f_off = 0, m_off = 0, shift = 0, bilinear = 0;
    };
    struct LayerRef {
        int cin, cout, k, stride, elu;
        size_t w_off, p_off, wino_off, w16_off, w4_off, sc_off, w4h_off = ~(size_t)0, d3h_off = ~(size_t)0, t3h_off = ~(size_t)0;
    };

    // One BasicConv.  srcs = {tensor id, shift}; out tensor must already exist.
    void conv(const std::string &path, std::vector<std::pair<int, int>> srcs, int out_t, int mul_t = -1,
              int res_t = -1)
    {
        const Arch &A = arch(u->layout);
        const LayerInfo &L = A.layers[A.find(path)];
        emit(path, LayerRef{L.cin, L.cout, L.k, L.stride, L.elu, L.w_off, L.p_off, L.wino_off, L.w16_off, L.w4_off, L.sc_off, L.w4h_off, L.d3h_off, L.t3h_off}, srcs, out_t, mul_t, res_t, 0,
             PreRef());
    }
    // A derived 1x1 layer (DerivedInfo): `linear` ones store the pre-activations [f | m] for a finer level to add,
    // gated ones are the AFF layer itself on the inputs of its own level, with the layer's own bias / BatchNorm.
    void conv_derived(const std::string &name, std::vector<std::pair<int, int>> srcs, int out_t, int linear, PreRef pre)
    {
        const Arch &A = arch(u->layout);
        const DerivedInfo &D = A.derived[A.find_derived(name)];
        const LayerInfo &P0 = A.layers[D.parts[0]];
        emit(name, LayerRef{D.cin, D.cout, 1, 1, P0.elu, D.w_off, linear ? D.p_off : P0.p_off, NO_WINO, NO_WINO, NO_WINO, NO_WINO, NO_WINO, D.h_off}, srcs, out_t, -1, -1,
             linear, pre);
    }

    void emit(const std::string &path, const LayerRef &L, const std::vector<std::pair<int, int>> &srcs, int out_t, int mul_t,
              int res_t, int linear, PreRef pre)
    {
        Op op;
        memset(&op.d, 0, sizeof(op.d));
        op.kind = Op::CONV;
        op.label = path;
        op.mul_t = mul_t;
        op.res_t = res_t;
        op.out_t = out_t;
        op.in_t = -1;
        op.pre_t = pre.t;
        for (int i = 0; i < READ_CONV_MAX_SRC; ++i) op.src_t[i] = -1;
        const Tensor &o = u->tensors[out_t];
        op.d.n_src = (int)srcs.size();
        int cin = 0;
        for (size_t i = 0; i < srcs.size(); ++i) {
            const Tensor &t = u->tensors[srcs[i].first];
            op.src_t[i] = srcs[i].first;
            op.d.src[i].data = t.p;
            op.d.src[i].C = t.C;
            op.d.src[i].srcH = t.H;
            op.d.src[i].srcW = t.W;
            op.d.src[i].shift = srcs[i].second;
            cin += t.C;
        }
        op.d.inH = o.H * L.stride;
        op.d.inW = o.W * L.stride;
        op.d.Cout = L.cout;
        op.d.ksize = L.k;
        op.d.stride = L.stride;
        op.d.elu = L.elu;
        op.d.wpacked = L.w_off != NO_WINO ? u->packed + L.w_off : nullptr;      // lean blob: absent where the F(4x4) kernel runs the layer
        op.d.params = u->packed + L.p_off;
        op.d.wpacked_wino = L.wino_off != NO_WINO ? u->packed + L.wino_off : nullptr;
        op.d.wpacked_w16 = L.w16_off != NO_WINO ? u->packed + L.w16_off : nullptr;
        op.d.wpacked_w4 = L.w4_off != NO_WINO ? u->packed + L.w4_off : nullptr;
        op.d.wpacked_sc = L.sc_off != NO_WINO ? u->packed + L.sc_off : nullptr;
        op.d.wpacked_w4h = L.w4h_off != NO_WINO ? u->packed + L.w4h_off : nullptr;
        op.d.wpacked_d3h = L.d3h_off != NO_WINO ? u->packed + L.d3h_off : nullptr;
        op.d.wpacked_t3h = L.t3h_off != NO_WINO ? u->packed + L.t3h_off : nullptr;
        op.d.mul = mul_t >= 0 ? u->tensors[mul_t].p : nullptr;
        op.d.residual = res_t >= 0 ? u->tensors[res_t].p : nullptr;
        op.d.out = o.p;
        op.d.out_cstride = o.C;
        op.d.config = -1;
        op.d.linear = linear;
        if (pre.t >= 0) {
            const Tensor &pt = u->tensors[pre.t];
            op.d.pre = pt.p;
            op.d.pre_cstride = pt.C;
            op.d.pre_f_off = pre.f_off;
            op.d.pre_m_off = pre.m_off;
            op.d.pre_shift = pre.shift;
            op.d.pre_bilinear = pre.bilinear;
            op.d.preH = pt.H;
            op.d.preW = pt.W;
        }
        op.flops = 2.0 * 2.0 * (double)o.H * o.W * L.cout * cin * L.k * L.k;
        op.is_c3s1 = (L.k == 3 && L.stride == 1 && L.cin == L.cout && L.cin >= BASE) ? 1 : 0;
        if (cin != L.cin) set_error("internal: layer %s expects Cin=%d, plan gives %d", path.c_str(), L.cin, cin);
        if (o.C != (linear ? 2 : 1) * L.cout && o.ext < 0)
            set_error("internal: layer %s writes %d channels into a %d-channel tensor", path.c_str(), (linear ? 2 : 1) * L.cout, o.C);
        op.lane = cur_lane;
        u->tensors[out_t].lane = cur_lane;
        if (cur_lane == 0) {
            for (size_t i = 0; i < srcs.size(); ++i)
                if (u->tensors[srcs[i].first].lane > 0) op.wait_mask |= 1 << (u->tensors[srcs[i].first].lane - 1);
            if (mul_t >= 0 && u->tensors[mul_t].lane > 0) op.wait_mask |= 1 << (u->tensors[mul_t].lane - 1);
            if (res_t >= 0 && u->tensors[res_t].lane > 0) op.wait_mask |= 1 << (u->tensors[res_t].lane - 1);
        }
        u->ops.push_back(op);
    }
    void up4(int in_t, int out_t)
    {
        Op op;
        memset(&op.d, 0, sizeof(op.d));
        op.kind = Op::UP4;
        op.label = "up4(" + u->tensors[in_t].name + ")";
        op.in_t = in_t;
        op.out_t = out_t;
        op.mul_t = op.res_t = -1;
        for (int i = 0; i < READ_CONV_MAX_SRC; ++i) op.src_t[i] = -1;
        op.flops = 0;
        op.is_c3s1 = 0;
        u->ops.push_back(op);
    }

    // 4 ResBlocks: x + BC1(BC0(x))  (unet.py:11-20); returns the tensor id holding the result.
    int resblocks(const std::string &prefix, int x, int level, int C)
    {
        int t = tensor(prefix + ".t", level, C);
        int y = tensor(prefix + ".y", level, C);
        int spare = tensor(prefix + ".y2", level, C);
        for (int j = 0; j < NUM_RES; ++j) {
            const std::string p = prefix + ".layers." + std::to_string(j) + ".main.";
            conv(p + "0", {{x, 0}}, t);
            conv(p + "1", {{t, 0}}, y, -1, x);
            // rotate: the block input may be needed later only for j == 0 (never overwritten:
            // `spare` takes its place in the ring)
            const int nx = y;
            y = (j == 0) ? spare : x;
            x = nx;
        }
        return x;
    }

    int scm(int n, int xin, int level)
    {
        const int P = (n == 0) ? BASE * 8 : (n == 1 ? BASE * 4 : BASE * 2);
        const std::string p = "SCM" + std::to_string(n);
        int a = tensor(p + ".a", level, P / 4);
        int b = tensor(p + ".b", level, P / 2);
        int c = tensor(p + ".c", level, P / 2);
        int d = tensor(p + ".d", level, P - IN_CH);
        int z = tensor(p + ".out", level, P);
        cur_lane = 1 + n;
        conv(p + ".main.0", {{xin, 0}}, a);
        conv(p + ".main.1", {{a, 0}}, b);
        conv(p + ".main.2", {{b, 0}}, c);
        conv(p + ".main.3", {{c, 0}}, d);
        conv(p + ".conv", {{xin, 0}, {d, 0}}, z);
        cur_lane = 0;
        return z;
    }

    void build()
    {
        u->tensors.clear();
        u->ops.clear();
        used = 0;
        const int x0 = external("x0", 0, 0, IN_CH), x1 = external("x1", 1, 1, IN_CH);
        const int x2 = external("x2", 2, 2, IN_CH), x3 = external("x3", 3, 3, IN_CH);
        const int rgb = external("rgb", 4, 0, 3);

        // unet.py:214-216
        const int z2 = scm(2, x1, 1), z4 = scm(1, x2, 2), z8 = scm(0, x3, 3);
        // unet.py:219-220
        int xf = tensor("fe0", 0, BASE);
        conv("feat_extract.0", {{x0, 0}}, xf);
        const int res1 = resblocks("Encoder.0", xf, 0, BASE);
        // unet.py:224-226: z = fe1(res1); z = z + merge(z*z2); res2 = E1(z)
        int f1 = tensor("fe1", 1, BASE * 2), fam2 = tensor("fam2", 1, BASE * 2);
        conv("feat_extract.1", {{res1, 0}}, f1);
        conv("FAM2.merge", {{f1, 0}}, fam2, z2, f1);
        const int res2 = resblocks("Encoder.1", fam2, 1, BASE * 2);
        // unet.py:228-230
        int f2 = tensor("fe2", 2, BASE * 4), fam1 = tensor("fam1", 2, BASE * 4);
        conv("feat_extract.2", {{res2, 0}}, f2);
        conv("FAM1.merge", {{f2, 0}}, fam1, z4, f2);
        const int res3 = resblocks("Encoder.2", fam1, 2, BASE * 4);
        // unet.py:232-235
        int f6 = tensor("fe6", 3, BASE * 8), fam0 = tensor("fam0", 3, BASE * 8);
        conv("feat_extract.6", {{res3, 0}}, f6);
        conv("FAM0.merge", {{f6, 0}}, fam0, z8, f6);
        const int zb = resblocks("Encoder.3", fam0, 3, BASE * 8);

        // unet.py:239-254: nearest resamples folded into the AFF 1x1 convs.
        // shift > 0: source is finer than the destination (down-sampling), < 0: coarser.
        int a0 = tensor("aff0.t", 0, BASE), r1 = tensor("aff0.out", 0, BASE);
        int a1 = tensor("aff1.t", 1, BASE * 2), r2 = tensor("aff1.out", 1, BASE * 2);
        int a2 = tensor("aff2.t", 2, BASE * 4), r3 = tensor("aff2.out", 2, BASE * 4);
        if (g_unet_aff_split) {
            // Every AFF input that lives at a coarser level is multiplied by its weight block AT that level (linear
            // launches q3 -> q2 -> q1, each adding the up-sampled sum of the coarser ones), and the gated layer at the
            // AFF's own level adds the result to its pre-activations: 11.2 instead of 46 GFLOP, same sums re-associated.
            //   q3 = [f0 f1 f2 | m0 m1 m2] (z), q2 = [f0 f1 | m0 m1] (res3 + up q3), q1 = [f0 | m0] (res2 + up q2)
            const int q3 = tensor("aff.q3", 3, 2 * BASE * 7), q2 = tensor("aff.q2", 2, 2 * BASE * 3);
            const int q1 = tensor("aff.q1", 1, 2 * BASE);
            conv_derived("AFFq3", {{zb, 0}}, q3, 1, PreRef());
            conv_derived("AFFq2", {{res3, 0}}, q2, 1, PreRef{q3, 0, BASE * 7, 1});
            conv_derived("AFFs.2.conv.0r", {{res1, 2}, {res2, 1}, {res3, 0}}, a2, 0, PreRef{q3, BASE * 3, BASE * 10, 1});
            conv_derived("AFFq1", {{res2, 0}}, q1, 1, PreRef{q2, 0, BASE * 3, 1});
            conv_derived("AFFs.1.conv.0r", {{res1, 1}, {res2, 0}}, a1, 0, PreRef{q2, BASE, BASE * 4, 1});
            conv_derived("AFFs.0.conv.0r", {{res1, 0}}, a0, 0, PreRef{q1, 0, BASE, 1});
        } else {
            conv("AFFs.0.conv.0", {{res1, 0}, {res2, -1}, {res3, -2}, {zb, -3}}, a0);
            conv("AFFs.1.conv.0", {{res1, 1}, {res2, 0}, {res3, -1}, {zb, -2}}, a1);
            conv("AFFs.2.conv.0", {{res1, 2}, {res2, 1}, {res3, 0}, {zb, -1}}, a2);
        }
        conv("AFFs.0.conv.1", {{a0, 0}}, r1);
        conv("AFFs.1.conv.1", {{a1, 0}}, r2);
        conv("AFFs.2.conv.1", {{a2, 0}}, r3);

        // unet.py:257-265
        int z = resblocks("Decoder.0", zb, 3, BASE * 8);
        // Convs.k over cat[Upsample4(fe), r]: with unet_up_fold the up-sampled tensor is never written — `Convs.k.u` applies its
        // weight block to fe at fe's level (linear, [f | m]) and `Convs.k.r` adds the bilinear x4 of that to its pre-activations
        auto up_conv = [&](int k, int fe, int level, int C, int r, int out) {
            const std::string L = "Convs." + std::to_string(k);
            if (g_unet_up_fold) {
                const int q = tensor("convs" + std::to_string(k) + ".q", level + 2, 2 * C);
                conv_derived(L + ".u", {{fe, 0}}, q, 1, PreRef());
                conv_derived(L + ".r", {{r, 0}}, out, 0, PreRef{q, 0, C, 2, 1});
            } else {
                const int up = tensor("up" + std::to_string(k), level, C);
                up4(fe, up);
                conv(L, {{up, 0}, {r, 0}}, out);
            }
        };
        int f7 = tensor("fe7", 4, BASE * 4), c0 = tensor("convs0", 2, BASE * 4);
        conv("feat_extract.7", {{z, 0}}, f7);
        up_conv(0, f7, 2, BASE * 4, r3, c0);
        z = resblocks("Decoder.1", c0, 2, BASE * 4);
        // unet.py:268-273
        int f3 = tensor("fe3", 3, BASE * 2), c1 = tensor("convs1", 1, BASE * 2);
        conv("feat_extract.3", {{z, 0}}, f3);
        up_conv(1, f3, 1, BASE * 2, r2, c1);
        z = resblocks("Decoder.2", c1, 1, BASE * 2);
        // unet.py:276-282
        int f4 = tensor("fe4", 2, BASE), c2 = tensor("convs2", 0, BASE);
        conv("feat_extract.4", {{z, 0}}, f4);
        up_conv(2, f4, 0, BASE, r1, c2);
        z = resblocks("Decoder.3", c2, 0, BASE);
        conv("feat_extract.5", {{z, 0}}, rgb);
        u->ws_used = used;
    }
};

int check_hw(int H, int W)
{
    READ_CHECK_ARG(H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0,
                   "UNet viewport must be a positive multiple of 16 in both dimensions (got %dx%d)", W, H);
    READ_CHECK_ARG((long long)H * W * BASE * 15 < (1ll << 31), "viewport too large");
    return READ_OK;
}

int g_unet_streams = 0;       // read_tuning_set("unet_streams", 1): SCM chains on side streams (measured slower: 135.8 vs 141.7 frames/s)

int run(read_unet *u, const float *const ext[5], int rgb_cstride, hipStream_t s0, bool timed)
{
    int ev = 0;
    if (!timed && g_unet_streams && !u->side[0]) {
        for (int i = 0; i < 3; ++i) {
            READ_CHECK_HIP(hipStreamCreateWithFlags(&u->side[i], hipStreamNonBlocking));
            READ_CHECK_HIP(hipEventCreateWithFlags(&u->ev_done[i], hipEventDisableTiming));
        }
        READ_CHECK_HIP(hipEventCreateWithFlags(&u->ev_start, hipEventDisableTiming));
    }
    const bool fork = !timed && g_unet_streams && u->side[0];
    int last_of_lane[4] = {-1, -1, -1, -1};
    if (fork) {
        READ_CHECK_HIP(hipEventRecord(u->ev_start, s0));          // inputs (and the previous frame) are ordered before this
        for (int i = 0; i < 3; ++i) READ_CHECK_HIP(hipStreamWaitEvent(u->side[i], u->ev_start, 0));
        for (size_t i = 0; i < u->ops.size(); ++i) last_of_lane[u->ops[i].lane] = (int)i;
    }
    int waited = 0;
    for (size_t oi = 0; oi < u->ops.size(); ++oi) {
        Op &op = u->ops[oi];
        hipStream_t s = (fork && op.lane > 0) ? u->side[op.lane - 1] : s0;
        if (fork && (op.wait_mask & ~waited)) {
            for (int i = 0; i < 3; ++i)
                if ((op.wait_mask & ~waited) & (1 << i)) READ_CHECK_HIP(hipStreamWaitEvent(s0, u->ev_done[i], 0));
            waited |= op.wait_mask;
        }
        if (timed) READ_CHECK_HIP(hipEventRecord(u->events[ev++], s));
        auto ptr = [&](int t) -> const float * {
            if (t < 0) return nullptr;
            const Tensor &T = u->tensors[t];
            return T.ext >= 0 ? ext[T.ext] : T.p;
        };
        if (op.kind == Op::CONV) {
            read_conv_desc d = op.d;
            for (int i = 0; i < d.n_src; ++i) d.src[i].data = ptr(op.src_t[i]);
            d.mul = ptr(op.mul_t);
            d.residual = ptr(op.res_t);
            d.pre = ptr(op.pre_t);
            d.out = const_cast<float *>(ptr(op.out_t));
            if (u->tensors[op.out_t].ext == 4) {
                d.out_cstride = rgb_cstride;
                d.fill_pad = rgb_cstride > d.Cout;   // RGBA: alpha = 1 (READ/gl/nn.py:124)
                d.out_fill = 1.0f;
            }
            const int rc = launch_gated_conv(&d, s);
            if (rc != READ_OK) return rc;
        } else {
            const Tensor &I = u->tensors[op.in_t];
            const int rc = read_bilinear_up4(ptr(op.in_t), I.H, I.W, I.C, const_cast<float *>(ptr(op.out_t)), s);
            if (rc != READ_OK) return rc;
        }
        if (fork && op.lane > 0 && (int)oi == last_of_lane[op.lane])
            READ_CHECK_HIP(hipEventRecord(u->ev_done[op.lane - 1], s));
    }
    if (fork)                                                       // a lane nobody consumed still joins the caller's stream
        for (int i = 0; i < 3; ++i)
            if (!(waited & (1 << i)) && last_of_lane[i + 1] >= 0) READ_CHECK_HIP(hipStreamWaitEvent(s0, u->ev_done[i], 0));
    if (timed) READ_CHECK_HIP(hipEventRecord(u->events[ev++], s0));
    return READ_OK;
}

}  // namespace

extern "C" int read_unet_layer_count(void) { return (int)arch().layers.size(); }

extern "C" int read_unet_layer_info(int i, const char **path, int *cin, int *cout, int *ksize, int *stride, int *elu)
{
    const Arch &A = arch();
    READ_CHECK_ARG(i >= 0 && i < (int)A.layers.size(), "read_unet_layer_info: index %d out of range", i);
    const LayerInfo &L = A.layers[i];
    if (path) *path = L.path.c_str();
    if (cin) *cin = L.cin;
    if (cout) *cout = L.cout;
    if (ksize) *ksize = L.k;
    if (stride) *stride = L.stride;
    if (elu) *elu = L.elu;
    return READ_OK;
}

extern "C" size_t read_unet_raw_floats(void) { return arch().raw_floats; }
extern "C" size_t read_unet_packed_floats(void) { return arch().packed_floats; }
extern "C" size_t read_unet_packed_floats_layout(int layout)
{
    return (layout == READ_UNET_LAYOUT_FULL || layout == READ_UNET_LAYOUT_LEAN) ? arch(layout).packed_floats : 0;
}

extern "C" int read_unet_pack_host(const float *raw, float bn_eps, float *packed)
{
    return read_unet_pack_host_layout(raw, bn_eps, packed, READ_UNET_LAYOUT_FULL);
}

extern "C" int read_unet_pack_host_layout(const float *raw, float bn_eps, float *packed, int layout)
{
    READ_CHECK_ARG(raw && packed, "read_unet_pack_host: null pointer");
    READ_CHECK_ARG(layout == READ_UNET_LAYOUT_FULL || layout == READ_UNET_LAYOUT_LEAN, "read_unet_pack_host: unknown layout %d", layout);
    for (const LayerInfo &L : arch(layout).layers) {
        const size_t wn = (size_t)L.cout * L.cin * L.k * L.k;
        const float *wf = raw + L.raw_off, *bf = wf + wn, *wm = bf + L.cout, *bm = wm + wn;
        const float *gamma = bm + L.cout, *beta = gamma + L.cout, *mean = beta + L.cout, *var = mean + L.cout;
        int rc = read_conv_pack_params_host(L.cout, bf, bm, gamma, beta, mean, var, bn_eps, packed + L.p_off);
        if (rc) return rc;
        if (L.w_off != NO_WINO) {
            rc = read_conv_pack_weights_host(L.cin, L.cout, L.k, L.kc, wf, wm, packed + L.w_off);
            if (rc) return rc;
        }
        if (L.wino_off != NO_WINO) {
            rc = read_conv_pack_wino_host(L.cin, L.cout, wf, wm, packed + L.wino_off);
            if (rc) return rc;
            rc = read_conv_pack_w16_host(L.cin, L.cout, wf, wm, packed + L.w16_off);
            if (rc) return rc;
        }
        if (L.w4_off != NO_WINO) {
            rc = read_conv_pack_w4_host(L.cin, L.cout, wf, wm, packed + L.w4_off);
            if (rc) return rc;
        }
        if (L.sc_off != NO_WINO) {
            rc = read_conv_pack_sc_host(L.cin, L.cout, wf, wm, packed + L.sc_off);
            if (rc) return rc;
        }
        if (L.w4h_off != NO_WINO) {
            rc = read_conv_pack_w4h_host(L.cin, L.cout, wf, wm, packed + L.w4h_off);
            if (rc) return rc;
        }
        if (L.d3h_off != NO_WINO) {
            rc = read_conv_pack_dkh_host(L.cin, L.cout, L.k, wf, wm, packed + L.d3h_off);
            if (rc) return rc;
        }
        if (L.t3h_off != NO_WINO) {
            rc = read_conv_pack_t3h_host(L.cin, L.cout, wf, wm, packed + L.t3h_off);
            if (rc) return rc;
        }
    }
    const Arch &A = arch(layout);
    for (const DerivedInfo &D : A.derived) {
        std::vector<float> wf((size_t)D.cout * D.cin), wm(wf.size()), zero(D.cout, 0.0f), one(D.cout, 1.0f);
        int co0 = 0;
        for (int li : D.parts) {
            const LayerInfo &L = A.layers[li];                     // 1x1: weights (cout, cin)
            const float *lf = raw + L.raw_off, *lm = lf + (size_t)L.cout * L.cin + L.cout;
            for (int co = 0; co < L.cout; ++co)
                for (int ci = 0; ci < D.cin; ++ci) {
                    wf[(size_t)(co0 + co) * D.cin + ci] = lf[(size_t)co * L.cin + D.ci0 + ci];
                    wm[(size_t)(co0 + co) * D.cin + ci] = lm[(size_t)co * L.cin + D.ci0 + ci];
                }
            co0 += L.cout;
        }
        int rc = read_conv_pack_weights_host(D.cin, D.cout, 1, 16, wf.data(), wm.data(), packed + D.w_off);
        if (rc) return rc;
        if (D.h_off != NO_WINO) {
            rc = read_conv_pack_dkh_host(D.cin, D.cout, 1, wf.data(), wm.data(), packed + D.h_off);
            if (rc) return rc;
        }
        rc = read_conv_pack_params_host(D.cout, zero.data(), zero.data(), one.data(), zero.data(), zero.data(), one.data(), 0.0f,
                                        packed + D.p_off);
        if (rc) return rc;
    }
    return READ_OK;
}

extern "C" size_t read_unet_workspace_bytes(int H, int W)
{
    if (check_hw(H, W) != READ_OK) return 0;
    read_unet tmp;
    tmp.H = H;
    tmp.W = W;
    tmp.packed = nullptr;
    tmp.ws = nullptr;
    Builder b{&tmp, true};
    b.build();
    return tmp.ws_used;
}

extern "C" int read_unet_create(read_unet_t **out, const float *packed, int H, int W, void *ws, size_t ws_bytes)
{
    return read_unet_create_layout(out, packed, H, W, ws, ws_bytes, READ_UNET_LAYOUT_FULL);
}

extern "C" int read_unet_create_layout(read_unet_t **out, const float *packed, int H, int W, void *ws, size_t ws_bytes, int layout)
{
    READ_CHECK_ARG(out && packed && ws, "read_unet_create: null pointer");
    READ_CHECK_ARG(layout == READ_UNET_LAYOUT_FULL || layout == READ_UNET_LAYOUT_LEAN, "read_unet_create: unknown layout %d", layout);
    READ_CHECK_ARG((uintptr_t)packed % 16 == 0 && (uintptr_t)ws % 256 == 0, "read_unet_create: misaligned weights/workspace");
    int rc = check_hw(H, W);
    if (rc) return rc;
    const size_t need = read_unet_workspace_bytes(H, W);
    if (ws_bytes < need) {
        set_error("read_unet_create: workspace %zu < %zu bytes", ws_bytes, need);
        return READ_ENOMEM;
    }
    read_unet *u = new read_unet();
    u->H = H;
    u->W = W;
    u->layout = layout;
    u->packed = packed;
    u->ws = (char *)ws;
    u->ws_bytes = ws_bytes;
    set_error("");
    Builder b{u, false};
    b.build();
    // a lean blob serves exactly the launches the F(4x4) kernel takes under the CURRENT tuning state and at THIS size: a knob
    // that sends such a layer elsewhere (conv_w4), or a tensor of 2 GiB and more, needs the full layout
    for (const Op &op : u->ops)
        if (op.kind == Op::CONV && !op.d.wpacked && !conv_uses_w4(&op.d) && !conv_uses_w4h(&op.d) && !conv_uses_d3h(&op.d) && !conv_uses_d3h_s2(&op.d))
            set_error("read_unet_create: layer %s is not run by the F(4x4) kernel here and the lean blob carries no other fragment "
                      "order for it (pack with READ_UNET_LAYOUT_FULL)", op.label.c_str());
    if (read_last_error()[0]) {   // the builder reports plan inconsistencies through set_error
        delete u;
        return READ_EINVAL;
    }
    *out = u;                     // (side streams are created on first use: the plan can be built without a device)
    return READ_OK;
}

extern "C" void read_unet_destroy(read_unet_t *u)
{
    if (!u) return;
    for (int i = 0; i < u->n_events; ++i) (void)hipEventDestroy(u->events[i]);
    delete[] u->events;
    for (int i = 0; i < 3; ++i) {
        if (u->side[i]) {
            (void)hipStreamSynchronize(u->side[i]);
            (void)hipStreamDestroy(u->side[i]);
        }
        if (u->ev_done[i]) (void)hipEventDestroy(u->ev_done[i]);
    }
    if (u->ev_start) (void)hipEventDestroy(u->ev_start);
    delete u;
}

extern "C" int read_unet_forward(read_unet_t *u, const float *x0, const float *x1, const float *x2,
                                 const float *x3, float *rgb, int rgb_cstride, void *stream)
{
    READ_CHECK_ARG(u && x0 && x1 && x2 && x3 && rgb, "read_unet_forward: null pointer");
    READ_CHECK_ARG(rgb_cstride == 3 || rgb_cstride == 4, "read_unet_forward: rgb_cstride must be 3 or 4");
    const float *ext[5] = {x0, x1, x2, x3, rgb};
    return run(u, ext, rgb_cstride, as_stream(stream), false);
}

extern "C" int read_unet_launch_count(read_unet_t *u) { return u ? (int)u->ops.size() : 0; }

extern "C" int read_unet_profile(read_unet_t *u, const float *x0, const float *x1, const float *x2,
                                 const float *x3, float *rgb, int rgb_cstride, void *stream, float *ms,
                                 double *flops, int *is_conv3x3_s1)
{
    READ_CHECK_ARG(u && x0 && x1 && x2 && x3 && rgb && ms, "read_unet_profile: null pointer");
    READ_CHECK_ARG(rgb_cstride == 3 || rgb_cstride == 4, "read_unet_profile: rgb_cstride must be 3 or 4");
    const int n = (int)u->ops.size();
    if (!u->events) {
        u->events = new hipEvent_t[n + 1];
        for (int i = 0; i <= n; ++i) READ_CHECK_HIP(hipEventCreate(&u->events[i]));
        u->n_events = n + 1;
    }
    const float *ext[5] = {x0, x1, x2, x3, rgb};
    hipStream_t s = as_stream(stream);
    const int rc = run(u, ext, rgb_cstride, s, true);
    if (rc) return rc;
    READ_CHECK_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        READ_CHECK_HIP(hipEventElapsedTime(&ms[i], u->events[i], u->events[i + 1]));
        if (flops) flops[i] = u->ops[i].flops;
        // 0: other, 1: 3x3/s1 C->C direct, 2: the same through the Winograd kernel (2.25x fewer MFMA flops than `flops`)
        if (is_conv3x3_s1)
            // 0: not in the 3x3/s1 C->C family; 1: direct kernel; 2: Winograd F(2x2,3x3); 4: Winograd F(4x4,3x3)
            //    5: Winograd F(4x4,3x3) with split operands on the f16 matrix cores
            //    6: direct 3x3 with split operands on the f16 matrix cores
            is_conv3x3_s1[i] = u->ops[i].is_c3s1 ? (u->ops[i].kind != Op::CONV ? 1 : conv_uses_d3h(&u->ops[i].d) ? 6 : conv_uses_w4h(&u->ops[i].d) ? 5 : conv_uses_w4(&u->ops[i].d) ? 4 :
                                                     conv_uses_wino(&u->ops[i].d) ? 2 : 1) : 0;
    }
    return READ_OK;
}

extern "C" int read_unet_launch_info(read_unet_t *u, int i, double *flops, int *is_conv3x3_s1, int *outH, int *outW,
                                     int *cin, int *cout)
{
    READ_CHECK_ARG(u && i >= 0 && i < (int)u->ops.size(), "read_unet_launch_info: bad handle or index");
    const Op &op = u->ops[i];
    const Tensor &o = u->tensors[op.out_t];
    if (flops) *flops = op.flops;
    if (is_conv3x3_s1) *is_conv3x3_s1 = op.is_c3s1;
    if (outH) *outH = o.H;
    if (outW) *outW = o.W;
    int ci = 0;
    for (int k = 0; k < READ_CONV_MAX_SRC; ++k)
        if (op.src_t[k] >= 0) ci += u->tensors[op.src_t[k]].C;
    if (op.kind == Op::UP4) ci = u->tensors[op.in_t].C;
    if (cin) *cin = ci;
    if (cout) *cout = o.C;
    return READ_OK;
}

extern "C" const char *read_unet_launch_label(read_unet_t *u, int i)
{
    return (u && i >= 0 && i < (int)u->ops.size()) ? u->ops[i].label.c_str() : "";
}

extern "C" const float *read_unet_debug_tensor(read_unet_t *u, const char *name, int *H, int *W, int *C)
{
    if (!u || !name) return nullptr;
    for (const Tensor &t : u->tensors)
        if (t.name == name) {
            if (H) *H = t.H;
            if (W) *W = t.W;
            if (C) *C = t.C;
            return t.p;
        }
    return nullptr;
}

namespace readhip {
void unet_set_streams(int v) { g_unet_streams = v; }
void unet_set_aff_split(int v) { g_unet_aff_split = v; }
void unet_set_up_fold(int v) { g_unet_up_fold = v; }
int unet_get(const char *key, int *value)
{
    if (!strcmp(key, "unet_streams")) *value = g_unet_streams;
    else if (!strcmp(key, "unet_aff_split")) *value = g_unet_aff_split;
    else if (!strcmp(key, "unet_up_fold")) *value = g_unet_up_fold;
    else return 0;
    return 1;
}
}
