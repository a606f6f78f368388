// a2: stacked-hourglass engine -- layer plan, parameter manifest, workspace planning and launches.
//
// The engine owns WHICH kernel runs on WHICH tensor (the plan below is the 2-stack, depth-4,
// pre-activation-bottleneck hourglass df2d uses: SURVEY.md App. B; constants reference df3d/config.py:18,33,36)
// and nothing else: weights, activations and the stream belong to the caller.
//
// BatchNorm handling (eval mode): a BN that directly follows a convolution (bn2, bn3 inside a bottleneck, the
// stem's bn1, the BN of fc) is folded into that convolution's weights and bias by the HOST packer; the BN on a
// bottleneck's *input* (bn1) cannot be folded because the raw tensor also feeds the skip connection, so it is
// applied as x*scale+shift -> ReLU while the conv1 kernel stages its input tile.
#include <algorithm>
#include <string>
#include <vector>

#include "common.h"
#include "hg_kernels.h"
#include "hg_bt_ring.h"
#include "hg_bt_l1.h"
#include "hg_head.h"
#include "hg_bt_ring_f32.h"
#include "hg_c1_f32.h"
#include "hg_l1_f32.h"
#include "hg_bt_wino_f32.h"
#include "hg_l1_wino_f32.h"
#include "hg_c1_res_f32.h"

using namespace hgk;

namespace {

enum StepKind { ST_STEM, ST_CONV, ST_POOL, ST_UPADD, ST_BOTTLENECK, ST_HEAD };

struct TensorDesc {
    size_t off;  // elements per view, from the start of the activation area
    int h, w, c, pitch;
};

constexpr size_t VIRTUAL_OFF = ~size_t(0);

struct ConvPlan {
    int taps, cin, cout, cin_pad, cout_pad;
    bool preact, relu, nchw_out;
    size_t w_off, b_off, s_off, t_off;  // float offsets into the blob (s/t only when preact)
};

struct Step {
    StepKind kind;
    std::string name;
    int in, out, res;  // tensor ids (res = -1: none; UPADD: in = hi-res, res = low-res)
    ConvPlan conv;     // ST_CONV / ST_STEM; for ST_BOTTLENECK: conv = conv1, conv2b = conv2, conv3b = conv3
    ConvPlan conv2b, conv3b, conv4b;  // ST_HEAD: conv = fc, conv2b = score, conv3b = fc_, conv4b = score_
    bool last = false;
    int pool_out = -1;                // ST_BOTTLENECK: tensor receiving the fused 2x2 max-pool of `out`
    int in2 = -1;                     // ST_BOTTLENECK: low-resolution addend of the input (upsample + add fused on the consumer side)
    int add2 = -1;                    // ST_BOTTLENECK: low-resolution addend of the OUTPUT (upsample + add fused into the producer's epilogue)
    long long wstream = -1;           // ST_BOTTLENECK, bf16 256 -> 128 -> 128 -> 256: byte offset of its weight stream behind the bf16 blob
    int pool_in = -1;                 // ST_BOTTLENECK (ring kernels): tensor receiving the 2x2 max-pool of the block's INPUT
    long long wstream2 = -1;          // ST_HEAD (bf16, not last): byte offset of the phase-C weight stream
    bool l1 = false;                  // ST_BOTTLENECK, bf16 64 -> 64 -> 64 -> 128: hg_bt_l1.h (wstream = its LDS weight image)
    bool pool_only = false;           // ... whose full-resolution output nobody reads: `out` IS the pooled tensor
    double m1_elems = 0;              // activation elements per view this step moves in the fusion model M1
    int t1 = -1;                      // ST_BOTTLENECK, fp32 split form (hg_c1_f32.h): the tensor conv1's kernel writes and the tail kernel reads
    bool l2f = false;                 // ST_BOTTLENECK, fp32 layer2 128 -> 128 -> 128 -> 256 + skip convolution: conv1 + tail (hg_l1_f32.h)
    bool l1f = false;                 // ST_BOTTLENECK, fp32 layer1 64 -> 64 -> 64 -> 128 + skip convolution: conv1 + tail (hg_l1_f32.h)
    long long wstream_w2d = -1;       // ... 16-bit W2D: byte offset of the direct-load form of W2'
    long long wstream_c1 = -1;        // ... and the byte offset of conv1's weight stream
    int chain = -1;                   // >= 0: planned inside chain number `chain` (frees postponed: its tensors share no memory)
};

struct Allocator {
    // first-fit allocator over "elements per view"; offsets multiple of 64 elements
    struct Blk { size_t off, size; };
    std::vector<Blk> free_list;
    size_t top = 0, peak = 0;
    size_t alloc(size_t n) {
        n = (n + 63) & ~size_t(63);
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].size >= n) {
                size_t off = free_list[i].off;
                free_list[i].off += n;
                free_list[i].size -= n;
                if (!free_list[i].size) free_list.erase(free_list.begin() + i);
                return off;
            }
        size_t off = top;
        top += n;
        peak = std::max(peak, top);
        return off;
    }
    void release(size_t off, size_t n) {
        n = (n + 63) & ~size_t(63);
        free_list.push_back({off, n});
        std::sort(free_list.begin(), free_list.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < free_list.size();) {
            if (free_list[i].off + free_list[i].size == free_list[i + 1].off) {
                free_list[i].size += free_list[i + 1].size;
                free_list.erase(free_list.begin() + i + 1);
            } else
                ++i;
        }
        if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
            top = free_list.back().off;
            free_list.pop_back();
        }
    }
};

}  // namespace

struct df3d_hg {
    int dtype = DF3D_DTYPE_F32;
    int num_stacks = 2;
    int H = 256, W = 512;
    int classes = 19;
    int rb_override = 0;  // 0 = auto, 64 or 128: staged row bytes per K-step (tuning knob)
    int fuse = 1;         // 1 = 256->128->128->256 bottlenecks at >= 16x32 run as ONE fused kernel
    int fuse_upadd = 0;   // the hourglass' up1 + upsample(low3): 0 = a pass of its own (upadd_kernel); 1 (default) = added in the epilogue of
                          // the bottleneck that produces up1 (the low path runs first) wherever the level's input already has a pooled
                          // copy, consumer side otherwise; 2 = always folded into the input load of the consuming bottleneck (round 2)
    int l1 = 1;           // 1 = bf16 layer1 (64 -> 64 -> 64 -> 128) runs as the LDS-resident-weights kernel of hg_bt_l1.h
    int ring = 1;         // 1 = the 256 -> 128 -> 128 -> 256 bottlenecks take their weights through the LDS-DMA ring (hg_bt_ring*.h)
    int w2d = 1;          // 16-bit ring bottlenecks: 1 (default) = the 3x3's weights as direct per-wave fragment loads (hg_bt_ring.h W2D), bit-identical
    int ring2 = 1;        // 16-bit ring bottlenecks (with w2d): 1 (default) = round 4's form (hg_bt_ring.h MODE 2: phase 3 without DMA round trips on its
                          // path, streaming output stores); 0 = round 3's kernels (the A/B); bit-identical either way
    int split1 = 1;       // fp32: 1 (default) = plain 256 -> 128 -> 128 -> 256 blocks run as conv1 (every pixel once) + tail (hg_c1_f32.h), bit-identical
                          // (development: 8 + mask splits only the identity blocks (1), layer1 (2), layer2 (4))
    int wino = 1;         // exact-fp32 engine, split identity blocks: 1 (default) = the tail's 3x3 as Winograd F(2x2, 3x3) (hg_bt_wino_f32.h: 0.545 of the
                          // direct tail's MFMAs; fp32 tolerance against the oracle, NOT bit-identical to the direct kernels), 0 = direct implicit GEMM
    int c1res = 1;        // exact-fp32 engine with `wino`: 1 (default) = conv1 of the plain identity blocks with W1 resident in LDS (hg_c1_res_f32.h), bit-identical
                          // to conv1_ring_f32_kernel (0); may be switched between forwards (it changes neither the plan nor the weight streams)
    bool split_id() const { return split1 == 1 || (split1 >= 8 && (split1 & 1)); }
    bool split_l1() const { return split1 == 1 || (split1 >= 8 && (split1 & 2)); }
    bool split_l2() const { return split1 == 1 || (split1 >= 8 && (split1 & 4)); }
    bool uses_zero_page = false;
    size_t zero_off = 0;  // byte offset of 256 zero bytes behind the weight streams (split form: the 3x3 padding of the tail's LDS-DMA)
    int no_reuse = 0;     // 1 = the alias-free workspace plan: no tensor ever takes a released tensor's memory (tests: the default plan must match it bit for bit)
    int chain_views = 0;  // > 0: chains of full-resolution steps run in chunks of this many views (Infinity Cache residency); 0 = off
    std::vector<int> chain_end;   // step i starts a chain [i, chain_end[i]) (chain_end[i] = i + 1: no chain)
    size_t stream_bytes = 0;  // weight streams of all ring bottlenecks (behind the bf16 copy of the blob)
    hgk::StemU8 u8in{nullptr, nullptr, 0, 0, 0, {{0, 0, 0}, {1, 1, 1}, 0}};   // df3d_hg_forward_u8: the stem's input for the duration of that call
    std::vector<TensorDesc> tensors;
    std::vector<int> pooled_of;   // tensor id -> id of its max-pooled copy written by the producing fused bottleneck (-1: none)
    std::vector<Step> steps;
    std::vector<df3d_hg_param> params;
    size_t blob_floats = 0;
    size_t act_elems_per_view = 0;
    const float* blob = nullptr;
    const void* lowp = nullptr;  // bf16 copy of the blob (same offsets, in elements) when dtype = bf16
    int final_tensor = -1;
    double flops_per_view = 0, elems_per_view = 0;

    Allocator alloc;

    // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg)
    bool profiling = false;
    struct Timed { hipEvent_t a, b; int cls; double flops, bytes, bytes_m1, flops_executed; };
    std::vector<std::string> kernel_names;  // class id -> kernel instantiation name (as rocprofv3 prints it, shortened)
    int kernel_class(const std::string& name) {
        for (size_t i = 0; i < kernel_names.size(); ++i)
            if (kernel_names[i] == name) return (int)i;
        kernel_names.push_back(name);
        return (int)kernel_names.size() - 1;
    }
    std::vector<Timed> timed;
    std::vector<hipEvent_t> event_pool;

    bool lp() const { return dtype == DF3D_DTYPE_BF16 || dtype == DF3D_DTYPE_F16; }   // a 16-bit engine (bf16 or f16: same plan, same kernels, other element type)
    int elem_bytes() const { return lp() ? 2 : 4; }
    // byte offset of the weight streams in the caller's "lowp" buffer: behind the 16-bit copy of the blob (bf16 / f16), at its start (f32)
    // (f32s: behind the pre-split float32 copy of the blob, hg_kernels.h f32s_presplit_kernel)
    size_t stream_base() const { return lp() ? (blob_floats * 2 + 255) & ~size_t(255) : dtype == DF3D_DTYPE_F32S ? (blob_floats * 4 + 255) & ~size_t(255) : 0; }

    // every step-creating site brackets its accounting: m1_open() before the first elems_per_view update that belongs to the
    // step, push_step(), m1_close() after the last one -> Step::m1_elems
    double m1_mark = 0;
    void m1_open() { m1_mark = elems_per_view; }
    void push_step(const Step& st) {
        steps.push_back(st);
        steps.back().chain = defer_frees ? plan_chain : -1;
    }
    void m1_close() { steps.back().m1_elems = elems_per_view - m1_mark; }
    int new_tensor(int h, int w, int c, int pitch = 0) {
        if (!pitch) pitch = c;
        TensorDesc t{alloc.alloc((size_t)h * w * pitch), h, w, c, pitch};
        tensors.push_back(t);
        pooled_of.push_back(-1);
        return (int)tensors.size() - 1;
    }
    // a tensor that is never materialised (shape only): the full-resolution output of a pooled-output-only bottleneck
    int new_virtual_tensor(int h, int w, int c) {
        tensors.push_back(TensorDesc{VIRTUAL_OFF, h, w, c, c});
        pooled_of.push_back(-1);
        return (int)tensors.size() - 1;
    }
    // While a chain of full-resolution steps is being planned (see chain_end) frees are postponed to its end: the chain runs
    // chunk by chunk, so a tensor released inside it must not lend its memory to a later tensor of the same chain (whose slice
    // for chunk c could overlap the released tensor's slice for chunk c + 1, which is still to be written and read).
    bool defer_frees = false;
    int plan_chain = 0;
    std::vector<int> deferred;
    void free_tensor(int id) {
        if (defer_frees) {
            deferred.push_back(id);
            return;
        }
        const TensorDesc& t = tensors[id];
        if (no_reuse) return;
        if (t.off != VIRTUAL_OFF) alloc.release(t.off, (size_t)t.h * t.w * t.pitch);
    }
    void end_chain() {
        defer_frees = false;
        ++plan_chain;
        for (int id : deferred) free_tensor(id);
        deferred.clear();
    }
    size_t add_param(const std::string& name, int kind, int taps, int cin, int cout, int cin_pad, int cout_pad, size_t count,
                     int kperm = 0) {
        df3d_hg_param p;
        memset(&p, 0, sizeof(p));
        p.kperm = kperm;
        snprintf(p.name, sizeof(p.name), "%s", name.c_str());
        p.kind = kind;
        p.taps = taps;
        p.cin = cin;
        p.cout = cout;
        p.cin_pad = cin_pad;
        p.cout_pad = cout_pad;
        p.offset = blob_floats;
        p.count = count;
        blob_floats += (count + 63) & ~size_t(63);
        params.push_back(p);
        return p.offset;
    }

    ConvPlan plan_conv(const std::string& name, int taps, int cin, int cin_pad, int cout, bool preact, bool relu, bool nchw_out,
                       int kperm = 0) {
        const int cout_pad = (cout + 31) / 32 * 32;
        ConvPlan c{taps, cin, cout, cin_pad, cout_pad, preact, relu, nchw_out, 0, 0, 0, 0};
        c.w_off = add_param(name, 0, taps, cin, cout, cin_pad, cout_pad, (size_t)taps * cout_pad * cin_pad, kperm);
        c.b_off = add_param(name, 1, taps, cin, cout, cin_pad, cout_pad, cout_pad);
        if (preact) {
            c.s_off = add_param(name, 2, taps, cin, cout, cin_pad, cout_pad, cin_pad);
            c.t_off = add_param(name, 3, taps, cin, cout, cin_pad, cout_pad, cin_pad);
        }
        return c;
    }
    void account_conv(double px, int taps, int cin, int cout, bool res) {
        flops_per_view += 2.0 * px * taps * cin * cout;
        elems_per_view += px * (cin + cout + (res ? cout : 0));
    }
    // one convolution step; returns the output tensor id
    int conv(const std::string& name, int in, int taps, int cout, bool preact, bool relu, int res, bool nchw_out = false) {
        const TensorDesc ti = tensors[in];
        Step st;
        st.kind = ST_CONV;
        st.name = name;
        st.in = in;
        st.res = res;
        st.conv = plan_conv(name, taps, ti.c, ti.pitch, cout, preact, relu, nchw_out);
        st.out = nchw_out ? -1 : new_tensor(ti.h, ti.w, cout, st.conv.cout_pad);
        m1_open();
        push_step(st);
        account_conv((double)ti.h * ti.w, taps, ti.c, cout, res >= 0);
        m1_close();
        return st.out;
    }
    // x2 >= 0: the block's input is x + nearest-upsample(x2) (the sum an ST_UPADD step would have written into x)
    // only_pool: the caller reads nothing but the max-pooled copy of the output
    // pool_input: the caller also needs max-pool(x) and nobody has produced it: the bf16 ring kernel writes it on the side
    // (pooled_of[x] is set when that happened)
    // a2 >= 0 (identity-skip blocks the fused kernels take; see can_add2): the block writes out + nearest-upsample(a2) under the
    // step name `sum_name` (the tensor an ST_UPADD step would have made of `out`)
    bool can_add2(int x) const {
        const TensorDesc& t = tensors[x];
        return fuse && t.c == 256 && t.h % 8 == 0 && t.w % 16 == 0;
    }
    int bottleneck(const std::string& name, int x, int planes, bool want_pool = false, int x2 = -1, bool only_pool = false, bool pool_input = false,
                   int a2 = -1, const std::string& sum_name = std::string()) {
        const int cin = tensors[x].c, cout = 2 * planes;
        const TensorDesc tx = tensors[x];
        const bool shape_ok = (cin == 256 && planes == 128) || (cin == 128 && planes == 128) || (cin == 64 && planes == 64);
        const bool fused_here = fuse && shape_ok && tx.h % 8 == 0 && tx.w % 16 == 0;
        if (x2 >= 0 && !(fused_here && fuse_upadd && cin == 256 && planes == 128)) {
            upadd(name + ".upadd", x, x2);  // no fused consumer: materialise the sum in place
            x2 = -1;
        }
        if (fused_here) {
            // the whole block in one kernel (hg_kernels.h: bottleneck_kernel); algorithmic work is accounted
            // exactly as for the separate convolutions (model M1), although far fewer bytes really move
            const bool ds = cin != cout;
            m1_open();
            Step st;
            st.kind = ST_BOTTLENECK;
            st.name = name + ".conv3";
            st.in = x;
            st.in2 = x2;
            st.add2 = a2;
            if (a2 >= 0) st.name = sum_name;
            if (x2 >= 0 || a2 >= 0) elems_per_view += (double)tx.h * tx.w * cin * 2.25;  // model M1 still counts the upsample + add pass
            st.res = ds ? -1 : x;
            st.conv = plan_conv(name + ".conv1", 1, cin, cin, planes, true, true, false);
            st.conv2b = plan_conv(name + ".conv2", 9, planes, planes, planes, false, true, false);
            if (ds) st.conv4b = plan_conv(name + ".downsample.0", 1, cin, cin, cout, false, false, false);
            st.conv3b = plan_conv(name + ".conv3", 1, planes, planes, cout, false, false, false, lp() ? 1 : 0);
            if (ring && lp() && cin == 128 && planes == 128 && x2 < 0 && !want_pool) {
                // layer2: the same ring kernel with 128 input channels and the skip convolution as eight more stages
                st.wstream = (long long)stream_bytes;
                stream_bytes += (size_t)br_nstage(128, true) * BR_STAGE_BYTES;
                if (w2d) {
                    st.wstream_w2d = (long long)stream_bytes;
                    stream_bytes += (size_t)BR_W2D_BYTES;
                }
            }
            if (ring && cin == 256 && planes == 128) {   // weights through the LDS-DMA ring (hg_bt_ring.h, hg_bt_ring_f32.h)
                st.wstream = (long long)stream_bytes;
                stream_bytes += (size_t)(lp() ? BR_NSTAGE : BRF_NSTAGE) * BR_STAGE_BYTES;
                if (w2d && lp()) {
                    st.wstream_w2d = (long long)stream_bytes;
                    stream_bytes += (size_t)BR_W2D_BYTES;
                }
                if (pool_input && x2 < 0 && pooled_of[x] < 0) {
                    st.pool_in = new_tensor(tx.h / 2, tx.w / 2, cin);
                    pooled_of[x] = st.pool_in;
                    elems_per_view += (double)tx.h * tx.w * cin * 1.25;  // model M1 still counts the pooling pass
                }
                if (split_id() && !lp()) {   // fp32 split form: conv1 on every pixel once, the rest on tiles
                    st.wstream_c1 = (long long)stream_bytes;
                    stream_bytes += (size_t)C1_NSTAGE * BR_STAGE_BYTES;
                    st.t1 = new_tensor(tx.h, tx.w, planes);
                    if (wino && dtype == DF3D_DTYPE_F32) {   // the tail's 3x3 in the Winograd domain: U = G g G^T as per-wave MFMA fragments
                        st.wstream_w2d = (long long)stream_bytes;
                        stream_bytes += (size_t)WN_STREAM_BYTES + C1R_W_BYTES;   // U, then W3 with permuted rows, then W1 for the LDS-resident conv1 (hg_c1_res_f32.h)
                    }
                }
            }
            if (ring && split_l1() && !lp() && cin == 64 && planes == 64 && ds && x2 < 0 && a2 < 0 && tx.h % BT_TH == 0 && tx.w % BT_TW == 0) {
                // fp32 layer1 in the split form: conv1 on every pixel once, the rest on tiles (hg_l1_f32.h)
                st.l1f = true;
                st.wstream = (long long)stream_bytes;
                stream_bytes += (size_t)L1F_NSTAGE * BR_STAGE_BYTES;
                st.wstream_c1 = (long long)stream_bytes;
                stream_bytes += (size_t)(64 / 16) * BR_STAGE_BYTES;
                st.t1 = new_tensor(tx.h, tx.w, planes);
                if (wino && dtype == DF3D_DTYPE_F32 && tx.w % L1W_TW == 0) {   // its 3x3 in the Winograd domain (hg_l1_wino_f32.h: 8 x 32 tiles): U | W3 | Wd
                    st.wstream_w2d = (long long)stream_bytes;
                    stream_bytes += (size_t)L1W_STREAM_BYTES;
                }
                if (want_pool && only_pool) {   // round 5: as the 16-bit layer1 kernel, the tail writes the pooled tensor only (the full-resolution
                                                // output, 15 of the block's 37 GB per 896 views, has no other reader)
                    st.pool_only = true;
                    st.out = new_tensor(tx.h / 2, tx.w / 2, cout);
                    const int virt = new_virtual_tensor(tx.h, tx.w, cout);
                    pooled_of[virt] = st.out;
                    elems_per_view += (double)tx.h * tx.w * cout * 1.25;  // model M1 still counts the pooling pass
                    push_step(st);
                    const double px = (double)tx.h * tx.w;
                    account_conv(px, 1, cin, planes, false);
                    account_conv(px, 9, planes, planes, false);
                    account_conv(px, 1, cin, cout, false);
                    account_conv(px, 1, planes, cout, true);
                    m1_close();
                    free_tensor(st.t1);
                    return virt;
                }
            }
            if (ring && split_l2() && !lp() && cin == 128 && planes == 128 && ds && x2 < 0 && a2 < 0 && !want_pool && tx.h % BT_TH == 0 && tx.w % BT_TW == 0) {
                st.l2f = true;   // fp32 layer2, the same split form
                st.wstream = (long long)stream_bytes;
                stream_bytes += (size_t)L2F_NSTAGE * BR_STAGE_BYTES;
                st.wstream_c1 = (long long)stream_bytes;
                stream_bytes += (size_t)(128 / 16) * BR_STAGE_BYTES;
                st.t1 = new_tensor(tx.h, tx.w, planes);
                if (wino && dtype == DF3D_DTYPE_F32) {   // its 3x3 in the Winograd domain too (hg_bt_wino_f32.h, L2): U | W3 | Wd
                    st.wstream_w2d = (long long)stream_bytes;
                    stream_bytes += (size_t)WN_STREAM_BYTES_L2;
                }
            }
            if (l1 && lp() && cin == 64 && planes == 64 && tx.h % 16 == 0 && tx.w % 16 == 0) {
                st.l1 = true;   // all weights resident in LDS (hg_bt_l1.h)
                st.wstream = (long long)stream_bytes;
                stream_bytes += L1_W_BYTES;
                if (want_pool && only_pool) {   // the full-resolution tensor is never written
                    st.pool_only = true;
                    st.out = new_tensor(tx.h / 2, tx.w / 2, cout);
                    const int virt = new_virtual_tensor(tx.h, tx.w, cout);
                    pooled_of[virt] = st.out;
                    elems_per_view += (double)tx.h * tx.w * cout * 1.25;  // model M1 still counts the pooling pass
                    push_step(st);
                    const double px = (double)tx.h * tx.w;
                    account_conv(px, 1, cin, planes, false);
                    account_conv(px, 9, planes, planes, false);
                    account_conv(px, 1, cin, cout, false);
                    account_conv(px, 1, planes, cout, true);
                    m1_close();
                    return virt;
                }
            }
            st.out = new_tensor(tx.h, tx.w, cout);
            if (want_pool) {  // the consumer max-pools this tensor: the epilogue writes the pooled copy too (no pool step)
                st.pool_out = new_tensor(tx.h / 2, tx.w / 2, cout);
                pooled_of[st.out] = st.pool_out;
                elems_per_view += (double)tx.h * tx.w * cout * 1.25;  // model M1 still counts the pooling pass
            }
            push_step(st);
            const double px = (double)tx.h * tx.w;
            account_conv(px, 1, cin, planes, false);
            account_conv(px, 9, planes, planes, false);
            if (ds) account_conv(px, 1, cin, cout, false);
            account_conv(px, 1, planes, cout, true);
            m1_close();
            if (st.t1 >= 0) free_tensor(st.t1);
            return st.out;
        }
        int a = conv(name + ".conv1", x, 1, planes, true, true, -1);
        int b = conv(name + ".conv2", a, 9, planes, false, true, -1);
        free_tensor(a);
        int skip = x;
        if (cin != cout) skip = conv(name + ".downsample.0", x, 1, cout, false, false, -1);
        int o = conv(name + ".conv3", b, 1, cout, false, false, skip);
        free_tensor(b);
        if (skip != x) free_tensor(skip);
        return o;
    }
    int pool(const std::string& name, int x) {
        if (pooled_of[x] >= 0) return pooled_of[x];  // already produced by the fused bottleneck that wrote x
        const TensorDesc t = tensors[x];
        Step st;
        st.kind = ST_POOL;
        st.name = name;
        st.in = x;
        st.res = -1;
        st.out = new_tensor(t.h / 2, t.w / 2, t.c, t.pitch);
        m1_open();
        push_step(st);
        elems_per_view += (double)t.h * t.w * t.c * 1.25;
        m1_close();
        return st.out;
    }
    // hi += upsample(lo), in place on hi
    int upadd(const std::string& name, int hi, int lo) {
        const TensorDesc t = tensors[hi];
        Step st;
        st.kind = ST_UPADD;
        st.name = name;
        st.in = hi;
        st.res = lo;
        st.out = hi;
        m1_open();
        push_step(st);
        elems_per_view += (double)t.h * t.w * t.c * 2.25;
        m1_close();
        return hi;
    }
    // Returns the up-path tensor; *lazy_lo receives the low-path tensor whose upsampled copy still has to be added to it
    // (the consumer -- always a bottleneck -- adds it while loading its input), or -1 when the sum was materialised.
    int hourglass(const std::string& name, int n, int x, int planes, int* lazy_lo) {
        const std::string lv = name + "." + std::to_string(n - 1);
        if (fuse && fuse_upadd == 1 && can_add2(x) && pooled_of[x] >= 0) {
            // LOW PATH FIRST, then up1 = bottleneck(x) whose epilogue adds upsample(low3) and writes the level's sum: the consumer is
            // a plain block (no second operand in its input load).  Needs max-pool(x) before up1 runs, i.e. a producer that has
            // already written it (everywhere except the second stack's input, which the branch below handles as in round 2).
            int low = pool(lv + ".pool", x);   // = pooled_of[x]
            int low1 = bottleneck(lv + ".1.0", low, planes, n > 1);
            free_tensor(low);
            int low2, inner_lo = -1;
            if (n > 1)
                low2 = hourglass(name, n - 1, low1, planes, &inner_lo);
            else
                low2 = bottleneck(lv + ".3.0", low1, planes);
            free_tensor(low1);
            int low3 = bottleneck(lv + ".2.0", low2, planes, false, inner_lo);
            free_tensor(low2);
            if (inner_lo >= 0) free_tensor(inner_lo);
            if (n == 4) defer_frees = chain_views > 0;   // outermost level: this step, the stack's residual block and its head form a chain
            int sum = bottleneck(lv + ".0.0", x, planes, false, -1, false, false, low3, lv + ".upadd");
            free_tensor(low3);
            *lazy_lo = -1;
            return sum;
        }
        int up1 = bottleneck(lv + ".0.0", x, planes, false, -1, false, true);
        int low = pool(lv + ".pool", x);
        int low1 = bottleneck(lv + ".1.0", low, planes, n > 1);
        free_tensor(low);
        int low2, inner_lo = -1;
        if (n > 1)
            low2 = hourglass(name, n - 1, low1, planes, &inner_lo);
        else
            low2 = bottleneck(lv + ".3.0", low1, planes);
        free_tensor(low1);
        int low3 = bottleneck(lv + ".2.0", low2, planes, false, inner_lo);
        free_tensor(low2);
        if (inner_lo >= 0) free_tensor(inner_lo);
        if (fuse && fuse_upadd) {
            *lazy_lo = low3;
        } else {
            upadd(lv + ".upadd", up1, low3);
            free_tensor(low3);
            *lazy_lo = -1;
        }
        return up1;
    }

    void build() {
        tensors.clear();
        pooled_of.clear();
        steps.clear();
        params.clear();
        blob_floats = 0;
        stream_bytes = 0;
        alloc = Allocator();
        flops_per_view = elems_per_view = 0;
        deferred.clear();
        plan_chain = 0;
        // chains (chain_views > 0) postpone the frees inside them: a chain's tensors must not share memory.  Without chunking (the
        // default) every tensor is released where its last consumer has run -- round 3 deferred always and planned 65 MB per view in
        // fp32 where 48 suffice (58 against 43 GB for one 896-view step)
        defer_frees = chain_views > 0;   // stem .. layer3 form a chain
        // stem
        Step st;
        st.kind = ST_STEM;
        st.name = "conv1";
        st.in = -1;
        st.res = -1;
        st.conv = ConvPlan{49, 3, 64, 3, 64, false, true, false, 0, 0, 0, 0};
        st.conv.w_off = add_param("conv1", 0, 49, 3, 64, 3, 64, 64 * 184);  // [148][64] f32 used; slot sized for the bf16 [64][184] tile
        st.conv.b_off = add_param("conv1", 1, 49, 3, 64, 3, 64, 64);
        st.out = new_tensor(H / 2, W / 2, 64);
        m1_open();
        push_step(st);
        flops_per_view += 2.0 * (H / 2) * (W / 2) * 147 * 64;
        elems_per_view += (double)H * W * 3 + (double)(H / 2) * (W / 2) * 64;
        m1_close();
        int x = st.out;
        int l1 = bottleneck("layer1.0", x, 64, true, -1, true);
        free_tensor(x);
        int p1 = pool("maxpool", l1);
        free_tensor(l1);
        int l2 = bottleneck("layer2.0", p1, 128);
        free_tensor(p1);
        x = bottleneck("layer3.0", l2, 128, true);
        free_tensor(l2);
        end_chain();
        for (int s = 0; s < num_stacks; ++s) {
            const std::string S = std::to_string(s);
            int ylo = -1;
            int y = hourglass("hg." + S + ".hg", 4, x, 128, &ylo);
            int r = bottleneck("res." + S + ".0", y, 128, false, ylo);
            free_tensor(y);
            if (ylo >= 0) free_tensor(ylo);
            if (fuse) {
                // fc -> score -> (fc_, score_) + x in one kernel (hg_head.h: head_kernel)
                const bool last = s == num_stacks - 1;
                const int kp = lp() ? 1 : 0;
                const TensorDesc tr = tensors[r];
                Step st;
                st.kind = ST_HEAD;
                st.last = last;
                if (ring) {   // Wfc through the LDS-DMA stage ring (hg_head.h)
                    st.wstream = (long long)stream_bytes;
                    stream_bytes += (size_t)(lp() ? HD_FC_STAGES : HD_FC_STAGES_F32) * BR_STAGE_BYTES;
                    if (!last && lp()) {
                        st.wstream2 = (long long)stream_bytes;
                        stream_bytes += (size_t)HD_FC2_STAGES * BR_STAGE_BYTES;
                    }
                }
                st.name = last ? "score." + S : "score_." + S;
                st.in = r;
                st.res = last ? -1 : x;
                st.conv = plan_conv("fc." + S + ".0", 1, 256, 256, 256, false, true, false);
                st.conv2b = plan_conv("score." + S, 1, 256, 256, classes, false, false, last, kp);
                const double px = (double)tr.h * tr.w;
                m1_open();
                account_conv(px, 1, 256, 256, false);
                account_conv(px, 1, 256, classes, false);
                if (!last) {
                    st.conv3b = plan_conv("fc_." + S, 1, 256, 256, 256, false, false, false, kp);
                    st.conv4b = plan_conv("score_." + S, 1, classes, 32, 256, false, false, false, kp);
                    account_conv(px, 1, 256, 256, true);
                    account_conv(px, 1, classes, 256, true);
                    st.out = new_tensor(tr.h, tr.w, 256);
                } else {
                    st.out = -1;
                }
                push_step(st);
                m1_close();
                free_tensor(r);
                free_tensor(x);
                end_chain();
                x = st.out;
                continue;
            }
            int f = conv("fc." + S + ".0", r, 1, 256, false, true, -1);
            free_tensor(r);
            const bool last = s == num_stacks - 1;
            int sc = conv("score." + S, f, 1, classes, false, false, -1, last);
            if (!last) {
                int t = conv("fc_." + S, f, 1, 256, false, false, x);
                free_tensor(f);
                free_tensor(x);
                int xn = conv("score_." + S, sc, 1, 256, false, false, t);
                free_tensor(sc);
                free_tensor(t);
                x = xn;
            } else {
                free_tensor(f);
                free_tensor(x);
            }
            end_chain();
        }
        act_elems_per_view = alloc.peak;
        zero_off = stream_bytes;
        uses_zero_page = false;
        for (const Step& st : steps) uses_zero_page = uses_zero_page || st.t1 >= 0;
        if (uses_zero_page) stream_bytes += 256;
        // chains: maximal runs of consecutive steps, each of which reads only the previous step's output (and tensors written
        // before the chain began) at the network's top resolutions
        chain_end.assign(steps.size(), 0);
        for (size_t i = 0; i < steps.size(); ++i) chain_end[i] = (int)i + 1;
        auto big = [&](const Step& st) {
            if (st.kind != ST_STEM && st.kind != ST_BOTTLENECK && st.kind != ST_HEAD) return false;
            const int ref = st.kind == ST_STEM ? st.out : st.in;
            return tensors[ref].h * 4 >= H / 4 * 4 && tensors[ref].h >= H / 4;   // 64 x 128 and above for the 256 x 512 input
        };
        for (size_t i = 0; i < steps.size();) {
            size_t j = i;
            if (big(steps[i])) {
                j = i + 1;
                while (j < steps.size() && big(steps[j]) && steps[j].in == steps[j - 1].out && steps[j].kind != ST_STEM && steps[i].chain >= 0 &&
                       steps[j].chain == steps[i].chain)
                    ++j;
                chain_end[i] = (int)j;
            }
            i = std::max(j, i + 1);
        }
    }
};

namespace {

// hipFuncSetAttribute acts on the CURRENT device: remember per device (bit i of `mask`) where it has been applied
// compute units of the current device (persistent kernels launch one workgroup per CU)
inline int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
    }
    return cached[dev];
}

inline bool first_use_on_this_device(unsigned& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) return true;
    const bool first = !(mask & (1u << dev));
    mask |= 1u << dev;
    return first;
}

template <typename T, int TAPS, int BN, int RB>
int launch_conv_t(const ConvArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * (BM + BN) * (RB + 16);
    static unsigned attr_done = 0;
    if (first_use_on_this_device(attr_done))
        DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<T, TAPS, BN, RB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const long long mt = (a.M + BM - 1) / BM;
    dim3 grid((unsigned)mt, (unsigned)(a.cout / BN));
    hipLaunchKernelGGL((conv_mfma_kernel<T, TAPS, BN, RB>), grid, dim3(256), LDS, s, a);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

template <typename T, int TAPS, int BN>
int launch_conv_rb(const ConvArgs& a, int rb, hipStream_t s) {
    if (rb == 128) return launch_conv_t<T, TAPS, BN, 128>(a, s);
    return launch_conv_t<T, TAPS, BN, 64>(a, s);
}

template <typename T>
int launch_conv(const ConvArgs& a, int taps, int rb, hipStream_t s) {
    int bn = (a.cout % 128 == 0) ? 128 : (a.cout % 64 == 0 ? 64 : 32);
    // a launch too small to fill the chip with 128-channel tiles (the 4 x 8 hourglass level: 224 workgroups for 896 views) takes
    // narrower ones: four times the workgroups, each with a quarter of the weights to pull -- the same K order per output, so the
    // same bits
    const long long wgs128 = ((a.M + BM - 1) / BM) * (a.cout / bn);
    if (bn == 128 && wgs128 < 2LL * cu_count()) bn = taps == 1 ? 32 : 64;
    if (taps == 1) {
        if (bn == 128) return launch_conv_rb<T, 1, 128>(a, rb, s);
        if (bn == 64) return launch_conv_rb<T, 1, 64>(a, rb, s);
        return launch_conv_rb<T, 1, 32>(a, rb, s);
    }
    if (bn == 128) return launch_conv_rb<T, 9, 128>(a, rb, s);
    if (bn == 64) return launch_conv_rb<T, 9, 64>(a, rb, s);
    df3d::set_error("3x3 convolution with cout %d unsupported", a.cout);
    return DF3D_EINVAL;
}


hipEvent_t get_event(df3d_hg* h) {
    if (!h->event_pool.empty()) {
        hipEvent_t e = h->event_pool.back();
        h->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ScopedTimer {
    df3d_hg* h;
    hipStream_t s;
    df3d_hg::Timed t;
    bool on;
    // bytes = the least this launch can move (inputs read once, outputs written once, intermediates on chip); bytes_m1 = what the
    // fusion model M1 of SURVEY.md 8(d) charges for the same work (every convolution's input and output, pooling and upsample passes)
    ScopedTimer(df3d_hg* h_, hipStream_t s_, const std::string& name, double flops, double bytes, double bytes_m1, double flops_executed = -1.0) : h(h_), s(s_), on(h_->profiling) {
        if (!on) return;
        t.a = get_event(h);
        t.b = get_event(h);
        t.cls = h->kernel_class(name);
        t.flops = flops;
        t.flops_executed = flops_executed >= 0.0 ? flops_executed : flops;
        t.bytes = bytes;
        t.bytes_m1 = bytes_m1;
        (void)hipEventRecord(t.a, s);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(t.b, s);
        h->timed.push_back(t);
    }
};

template <typename T, int CIN, int PL, bool DS, bool UP = false, bool ADD2 = false>
int launch_bottleneck_t(const BottleneckArgs& a, int blocks, hipStream_t s) {
    using C = BtCfg<T, CIN, PL, DS>;
    static unsigned attr_done = 0;
    if (first_use_on_this_device(attr_done))
        DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_kernel<T, CIN, PL, DS, UP, ADD2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL((bottleneck_kernel<T, CIN, PL, DS, UP, ADD2>), dim3(blocks), dim3(256), C::LDS_BYTES, s, a);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

template <typename T>
int launch_bottleneck(const BottleneckArgs& a, int cin, int pl, int blocks, hipStream_t s) {
    if (cin == 256 && pl == 128 && a.in2) return launch_bottleneck_t<T, 256, 128, false, true>(a, blocks, s);
    if (cin == 256 && pl == 128 && a.add2) return launch_bottleneck_t<T, 256, 128, false, false, true>(a, blocks, s);
    if (cin == 256 && pl == 128) return launch_bottleneck_t<T, 256, 128, false>(a, blocks, s);
    if (cin == 128 && pl == 128) return launch_bottleneck_t<T, 128, 128, true>(a, blocks, s);
    if (cin == 64 && pl == 64) return launch_bottleneck_t<T, 64, 64, true>(a, blocks, s);
    df3d::set_error("fused bottleneck %d -> %d unsupported", cin, pl);
    return DF3D_EINVAL;
}

template <typename T> struct TypeName;
template <> struct TypeName<float> { static constexpr const char* value = "float"; };
template <> struct TypeName<__hip_bfloat16> { static constexpr const char* value = "__hip_bfloat16"; };
template <> struct TypeName<_Float16> { static constexpr const char* value = "_Float16"; };
template <> struct TypeName<F32S> { static constexpr const char* value = "hgk::F32S"; };
// the element type of the kernels that only move or compare float32 data (pools, upsample-add, export): F32S tensors ARE float32 tensors
template <typename T> using StorageT = std::conditional_t<std::is_same<T, F32S>::value, float, T>;

// one launcher per 16-bit element type (hipFuncSetAttribute is per instantiation and per device)
template <typename T, bool UP, int CIN, bool ADD2, int MODE>
int launch_ring_lp_(const BtRingArgs& r, int blocks, int lds_bytes, hipStream_t s) {
    static unsigned attr_done = 0;
    if (first_use_on_this_device(attr_done))
        DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_ring_kernel<T, UP, CIN, ADD2, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL((bottleneck_ring_kernel<T, UP, CIN, ADD2, MODE>), dim3(blocks), dim3(256), lds_bytes, s, r);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}
// the ring kernel's MODE (hg_bt_ring.h) of a launch: 0 = all weights through the ring, 1 = W2D (round 3), 2 = W2D + the round-4 form (option `ring2`)
inline int ring_mode(const BtRingArgs& r, int ring2) { return !r.w2d ? 0 : ring2 ? 2 : 1; }
template <typename T, bool UP, int CIN, bool ADD2 = false>
int launch_ring_lp(const BtRingArgs& r, int ring2, int blocks, int lds_bytes, hipStream_t s) {
    const int mode = ring_mode(r, ring2);
    return mode == 2 ? launch_ring_lp_<T, UP, CIN, ADD2, 2>(r, blocks, lds_bytes, s)
           : mode    ? launch_ring_lp_<T, UP, CIN, ADD2, 1>(r, blocks, lds_bytes, s)
                     : launch_ring_lp_<T, UP, CIN, ADD2, 0>(r, blocks, lds_bytes, s);
}
template <typename T, bool UP, bool ADD2 = false, bool TAIL = false>
int launch_ring_f32(const BtRingArgs& r, int blocks, int lds_bytes, hipStream_t s) {
    static unsigned attr_done = 0;
    if (first_use_on_this_device(attr_done))
        DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_ring_f32_kernel<UP, ADD2, TAIL, T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL((bottleneck_ring_f32_kernel<UP, ADD2, TAIL, T>), dim3(blocks), dim3(256), lds_bytes, s, r);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

template <bool UP, bool ADD2, bool L2 = false>
int launch_wino_f32(const BtRingArgs& r, int blocks, hipStream_t s) {
    static unsigned attr_done = 0;
    if (first_use_on_this_device(attr_done))
        DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_wino_f32_kernel<UP, ADD2, L2>), hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES));
    // persistent: one workgroup per CU (it needs the whole register file), walking tiles with stride gridDim; a multiple of 8 keeps virtual
    // block ids on their XCD (hg_bt_wino_f32.h tile_of)
    const int cus = cu_count() & ~7;
    hipLaunchKernelGGL((bottleneck_wino_f32_kernel<UP, ADD2, L2>), dim3(blocks <= cus ? blocks : cus), dim3(256), WN_LDS_BYTES, s, r);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

template <typename T>
int run_steps(df3d_hg* h, const float* images_all, int n_all, int upto, float* heatmaps_all, unsigned char* act, hipStream_t s) {
    const int eb = sizeof(T);
    const char* const tname = TypeName<T>::value;   // as rocprofv3 prints the template argument
    // the weights the kernels read: the caller's float32 blob (f32), its 16-bit copy (bf16 / f16), its pre-split copy (f32s); biases and
    // BatchNorm coefficients always come from the blob
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(std::is_same<T, float>::value ? (const void*)h->blob : h->lowp);
    // one plan step on the views [v0, v0 + n) of the batch: every tensor is [views][h][w][pitch], so a view range is a
    // contiguous slice of each (the whole batch: v0 = 0, n = n_all)
    auto launch = [&](int i, int v0, int n) -> int {
        const Step& st = h->steps[i];
        auto tptr = [&](int id) -> unsigned char* {
            const TensorDesc& t = h->tensors[id];
            return act + (t.off * (size_t)n_all + (size_t)v0 * t.h * t.w * t.pitch) * eb;
        };
        const float* const images = images_all ? images_all + (size_t)v0 * h->H * h->W * 3 : nullptr;
        float* const heatmaps = heatmaps_all ? heatmaps_all + (size_t)v0 * h->classes * (h->H / 4) * (h->W / 4) : nullptr;
        switch (st.kind) {
            case ST_STEM: {
                StemArgs a;
                a.img = images;
                a.out = tptr(st.out);
                a.w = h->blob + st.conv.w_off;
                a.w_bf16 = eb == 2 ? wb + st.conv.w_off * eb : nullptr;
                a.bias = h->blob + st.conv.b_off;
                a.V = n;
                a.H = h->H;
                a.W = h->W;
                a.u8 = h->u8in;
                if (a.u8.frames) {
                    a.u8.frames += (size_t)v0 * a.u8.FH * a.u8.FW * a.u8.FC;
                    if (a.u8.flip) a.u8.flip += v0;
                }
                const int blocks = n * (h->H / 2 / 8) * (h->W / 2 / 16);
                const double opx = (double)n * (h->H / 2) * (h->W / 2);
                ScopedTimer tm(h, s, std::is_same<T, F32S>::value ? std::string("stem_f32s_kernel") : std::string(eb == 2 ? "stem_lp_kernel<" : "stem_kernel<") + TypeName<StorageT<T>>::value + ">",
                               2.0 * opx * 147 * 64, opx * (12.0 * 4 + 64.0 * eb), st.m1_elems * n * eb);
                if constexpr (sizeof(T) == 2) {
                    hipLaunchKernelGGL((stem_lp_kernel<T>), dim3(std::min(blocks, 4 * cu_count())), dim3(256), 0, s, a);   // persistent: weights once per workgroup
                } else if constexpr (std::is_same<T, F32S>::value) {
                    a.w_bf16 = wb + st.conv.w_off * eb;   // the hi / lo half tiles in the stem's slot of the pre-split copy (stem_relayout_f32s_kernel)
                    hipLaunchKernelGGL(stem_f32s_kernel, dim3(std::min(blocks, 2 * cu_count())), dim3(256), 0, s, a);   // persistent (57 KB of LDS: two per CU)
                } else
                    hipLaunchKernelGGL((stem_kernel<StorageT<T>>), dim3(std::min(blocks, 3 * cu_count())), dim3(256), 0, s, a);   // persistent: weights once per workgroup
                DF3D_LAUNCH_CHECK();
                break;
            }
            case ST_CONV: {
                const TensorDesc& ti = h->tensors[st.in];
                ConvArgs a;
                a.in = tptr(st.in);
                a.out = st.out >= 0 ? tptr(st.out) : nullptr;
                a.res = st.res >= 0 ? tptr(st.res) : nullptr;
                a.out_nchw = st.conv.nchw_out ? heatmaps : nullptr;
                a.w = wb + st.conv.w_off * eb;
                a.bias = h->blob + st.conv.b_off;
                a.scale = st.conv.preact ? h->blob + st.conv.s_off : nullptr;
                a.shift = st.conv.preact ? h->blob + st.conv.t_off : nullptr;
                a.M = (long long)n * ti.h * ti.w;
                a.H = ti.h;
                a.W = ti.w;
                a.cin = st.conv.cin_pad;
                a.cout = st.conv.cout_pad;
                a.in_pitch = ti.pitch;
                a.out_pitch = st.out >= 0 ? h->tensors[st.out].pitch : 0;
                a.res_pitch = st.res >= 0 ? h->tensors[st.res].pitch : 0;
                a.relu = st.conv.relu;
                a.cout_real = st.conv.cout;
                const int ke128 = 128 / eb;
                int rb = (st.conv.cin_pad % ke128 == 0) ? 128 : 64;
                if (h->rb_override == 64) rb = 64;
                const double mm = (double)a.M;
                int bn_tile = (a.cout % 128 == 0) ? 128 : (a.cout % 64 == 0 ? 64 : 32);
                if (bn_tile == 128 && ((a.M + BM - 1) / BM) * (a.cout / bn_tile) < 2LL * cu_count()) bn_tile = st.conv.taps == 1 ? 32 : 64;   // as launch_conv picks
                ScopedTimer tm(h, s, std::string("conv_mfma_kernel<") + tname + ", " + std::to_string(st.conv.taps) + ", " + std::to_string(bn_tile) + ", " + std::to_string(rb) + ">",
                               2.0 * mm * st.conv.taps * st.conv.cin * st.conv.cout,
                               mm * eb * (st.conv.cin + st.conv.cout + (st.res >= 0 ? st.conv.cout : 0)), st.m1_elems * n * eb);
                if (int rc = launch_conv<T>(a, st.conv.taps, rb, s)) return rc;
                break;
            }
            case ST_BOTTLENECK: {
                const TensorDesc& ti = h->tensors[st.in];
                const bool ds = st.res < 0;
                BottleneckArgs a;
                a.in = tptr(st.in);
                a.in2 = st.in2 >= 0 ? tptr(st.in2) : nullptr;
                a.add2 = st.add2 >= 0 ? tptr(st.add2) : nullptr;
                a.out = tptr(st.out);
                a.pool = st.pool_out >= 0 ? tptr(st.pool_out) : nullptr;
                a.w1 = wb + st.conv.w_off * eb;
                a.w2 = wb + st.conv2b.w_off * eb;
                a.w3 = wb + st.conv3b.w_off * eb;
                a.wd = ds ? wb + st.conv4b.w_off * eb : nullptr;
                a.b1 = h->blob + st.conv.b_off;
                a.b2 = h->blob + st.conv2b.b_off;
                a.b3 = h->blob + st.conv3b.b_off;
                a.bd = ds ? h->blob + st.conv4b.b_off : nullptr;
                a.s1 = h->blob + st.conv.s_off;
                a.t1 = h->blob + st.conv.t_off;
                a.V = n;
                a.H = ti.h;
                a.W = ti.w;
                const int cin = st.conv.cin, pl = st.conv.cout;
                const double px = (double)n * ti.h * ti.w;
                if (st.l1) {
                    BtL1Args r;
                    r.in = a.in;
                    r.out = st.pool_only ? nullptr : a.out;
                    r.pool = st.pool_only ? a.out : a.pool;
                    r.wimage = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream;
                    r.b1 = a.b1; r.b2 = a.b2; r.b3 = a.b3; r.bd = a.bd; r.s1 = a.s1; r.t1 = a.t1;
                    r.V = n; r.H = ti.h; r.W = ti.w;
                    ScopedTimer tm(h, s, std::string("bottleneck_l1_kernel<") + tname + ">", 2.0 * px * ((double)cin * pl + 9.0 * pl * pl + 2.0 * pl * pl + 2.0 * cin * pl),
                                   px * eb * (cin + (st.pool_only ? 0.5 * pl : 2.0 * pl)), st.m1_elems * n * eb);
                    const int tiles = n * (ti.h / L1_TH) * (ti.w / BT_TW);
                    if constexpr (sizeof(T) == 2) {
                        static unsigned attr_done = 0;
                        if (first_use_on_this_device(attr_done))
                            DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bottleneck_l1_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS_BYTES));
                        hipLaunchKernelGGL((bottleneck_l1_kernel<T>), dim3(std::min(tiles, cu_count())), dim3(L1_WAVES * 64), L1_LDS_BYTES, s, r);
                        DF3D_LAUNCH_CHECK();
                    }
                    break;
                }
                if (st.l2f) {
                    if constexpr (sizeof(T) == 4) {
                        const unsigned char* const sb = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base();
                        Conv1Args c;
                        c.in = a.in;
                        c.in2 = nullptr;
                        c.H = ti.h;
                        c.W = ti.w;
                        c.t1 = tptr(st.t1);
                        c.wstream = sb + st.wstream_c1;
                        c.b1 = a.b1; c.s1 = a.s1; c.t1c = a.t1;
                        c.M = (long long)n * ti.h * ti.w;
                        {
                            ScopedTimer tc(h, s, std::string("conv1_ring_f32_kernel<false, 128, 128, ") + tname + ">", 2.0 * px * cin * pl, px * 4.0 * (cin + pl), 0.0);
                            static unsigned attr_c1 = 0;
                            if (first_use_on_this_device(attr_c1))
                                DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_ring_f32_kernel<false, 128, 128, T>), hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_BYTES));
                            const unsigned c1_grid = (unsigned)std::min<long long>(c.M / 128, 2LL * cu_count());
                            hipLaunchKernelGGL((conv1_ring_f32_kernel<false, 128, 128, T>), dim3(c1_grid), dim3(256), C1_LDS_BYTES, s, c);
                            DF3D_LAUNCH_CHECK();
                        }
                        BtRingArgs r{};
                        r.in = a.in; r.out = a.out;
                        r.t1in = c.t1;
                        r.zeros = sb + h->zero_off;
                        r.wstream = sb + st.wstream;
                        r.b2 = a.b2; r.b3 = a.b3; r.bd = a.bd;
                        r.V = n; r.H = ti.h; r.W = ti.w;
                        if (std::is_same<T, float>::value && st.wstream_w2d >= 0) {   // option `wino`: layer2's 3x3 in the Winograd domain as well
                            r.w2d = sb + st.wstream_w2d;
                            ScopedTimer tw(h, s, "bottleneck_wino_f32_kernel<false, false, true>", 2.0 * px * (9.0 * pl * pl + (double)pl * 2 * pl + (double)cin * 2 * pl),
                                           px * 4.0 * (cin + pl + 2.0 * pl), st.m1_elems * n * eb, 2.0 * px * (4.0 * pl * pl + (double)pl * 2 * pl + (double)cin * 2 * pl));
                            if (int rc = launch_wino_f32<false, false, true>(r, n * (ti.h / BT_TH) * (ti.w / BT_TW), s)) return rc;
                            break;
                        }
                        ScopedTimer tm(h, s, std::string("layer2_tail_f32_kernel<") + tname + ">", 2.0 * px * (9.0 * pl * pl + (double)pl * 2 * pl + (double)cin * 2 * pl), px * 4.0 * (cin + pl + 2.0 * pl), st.m1_elems * n * eb);
                        static unsigned attr_t = 0;
                        if (first_use_on_this_device(attr_t))
                            DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(layer2_tail_f32_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, L2F_LDS_BYTES));
                        hipLaunchKernelGGL(layer2_tail_f32_kernel<T>, dim3(n * (ti.h / BT_TH) * (ti.w / BT_TW)), dim3(256), L2F_LDS_BYTES, s, r);
                        DF3D_LAUNCH_CHECK();
                    }
                    break;
                }
                if (st.l1f) {
                    if constexpr (sizeof(T) == 4) {
                        const unsigned char* const sb = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base();
                        Conv1Args c;
                        c.in = a.in;
                        c.in2 = nullptr;
                        c.H = ti.h;
                        c.W = ti.w;
                        c.t1 = tptr(st.t1);
                        c.wstream = sb + st.wstream_c1;
                        c.b1 = a.b1; c.s1 = a.s1; c.t1c = a.t1;
                        c.M = (long long)n * ti.h * ti.w;
                        {
                            ScopedTimer tc(h, s, std::string("conv1_ring_f32_kernel<false, 64, 64, ") + tname + ">", 2.0 * px * cin * pl, px * 4.0 * (cin + pl), 0.0);
                            static unsigned attr_c1 = 0;
                            if (first_use_on_this_device(attr_c1))
                                DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_ring_f32_kernel<false, 64, 64, T>), hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_BYTES));
                            const unsigned c1_grid = (unsigned)std::min<long long>(c.M / 128, 2LL * cu_count());
                            hipLaunchKernelGGL((conv1_ring_f32_kernel<false, 64, 64, T>), dim3(c1_grid), dim3(256), C1_LDS_BYTES, s, c);
                            DF3D_LAUNCH_CHECK();
                        }
                        BtRingArgs r{};
                        r.in = a.in;
                        r.out = st.pool_only ? nullptr : a.out;
                        r.pool = st.pool_only ? a.out : a.pool;
                        r.t1in = c.t1;
                        r.zeros = sb + h->zero_off;
                        r.wstream = sb + st.wstream;
                        r.b2 = a.b2; r.b3 = a.b3; r.bd = a.bd;
                        r.V = n; r.H = ti.h; r.W = ti.w;
                        if (std::is_same<T, float>::value && st.wstream_w2d >= 0) {   // option `wino`: layer1's 3x3 in the Winograd domain (8 x 32 tiles, persistent)
                            r.w2d = sb + st.wstream_w2d;
                            ScopedTimer tw(h, s, "layer1_wino_f32_kernel", 2.0 * px * (9.0 * pl * pl + (double)pl * 2 * pl + (double)cin * 2 * pl),
                                           px * 4.0 * (cin + pl + (st.pool_only ? 0.5 * pl : 2.0 * pl)), st.m1_elems * n * eb, 2.0 * px * (4.0 * pl * pl + (double)pl * 2 * pl + (double)cin * 2 * pl));
                            static unsigned attr_w = 0;
                            if (first_use_on_this_device(attr_w))
                                DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(layer1_wino_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L1W_LDS_BYTES));
                            const int tiles = n * (ti.h / BT_TH) * (ti.w / L1W_TW), cus = cu_count() & ~7;
                            hipLaunchKernelGGL(layer1_wino_f32_kernel, dim3(tiles <= cus ? tiles : cus), dim3(256), L1W_LDS_BYTES, s, r);
                            DF3D_LAUNCH_CHECK();
                            break;
                        }
                        ScopedTimer tm(h, s, std::string("layer1_tail_f32_kernel<") + tname + ">", 2.0 * px * (9.0 * pl * pl + (double)pl * 2 * pl + (double)cin * 2 * pl), px * 4.0 * (cin + pl + (st.pool_only ? 0.5 * pl : 2.0 * pl)), st.m1_elems * n * eb);
                        static unsigned attr_t = 0;
                        if (first_use_on_this_device(attr_t))
                            DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(layer1_tail_f32_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, L1F_LDS_BYTES));
                        hipLaunchKernelGGL(layer1_tail_f32_kernel<T>, dim3(n * (ti.h / BT_TH) * (ti.w / BT_TW)), dim3(256), L1F_LDS_BYTES, s, r);
                        DF3D_LAUNCH_CHECK();
                    }
                    break;
                }
                if (st.wstream >= 0) {
                    BtRingArgs r;
                    r.in = a.in; r.in2 = a.in2; r.add2 = a.add2; r.out = a.out; r.pool = a.pool;
                    r.t1in = nullptr; r.zeros = nullptr;
                    r.pool_in = st.pool_in >= 0 ? tptr(st.pool_in) : nullptr;
                    r.wstream = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream;
                    r.w2d = st.wstream_w2d >= 0 ? reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream_w2d : nullptr;
                    r.b1 = a.b1; r.b2 = a.b2; r.b3 = a.b3; r.bd = a.bd; r.s1 = a.s1; r.t1 = a.t1;
                    r.V = n; r.H = ti.h; r.W = ti.w;
                    if (ds) {   // 16-bit layer2
                        ScopedTimer tm(h, s, std::string("bottleneck_ring_kernel<") + tname + ", false, 128, false, " + std::to_string(ring_mode(r, h->ring2)) + ">", 2.0 * px * ((double)cin * pl + 9.0 * pl * pl + 2.0 * pl * pl + 2.0 * cin * pl), px * eb * (cin + 2.0 * pl), st.m1_elems * n * eb);
                        if constexpr (sizeof(T) == 2)
                            if (int rc = launch_ring_lp<T, false, 128>(r, h->ring2, n * (ti.h / BT_TH) * (ti.w / BT_TW), BR_LDS_BYTES, s)) return rc;
                        break;
                    }
                    const bool split = eb == 4 && st.t1 >= 0;
                    if (split) {   // fp32 split form: conv1 for every pixel of the level, then the tail on tiles
                        if constexpr (sizeof(T) == 4) {
                            Conv1Args c;
                            c.in = a.in;
                            c.in2 = a.in2;
                            c.H = ti.h;
                            c.W = ti.w;
                            c.t1 = tptr(st.t1);
                            c.wstream = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream_c1;
                            c.b1 = a.b1; c.s1 = a.s1; c.t1c = a.t1;
                            c.M = (long long)n * ti.h * ti.w;
                            r.t1in = c.t1;
                            r.zeros = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + h->zero_off;
                            const bool resident = std::is_same<T, float>::value && st.wstream_w2d >= 0 && !a.in2 && h->c1res;   // option `wino`: W1 resident in LDS
                            ScopedTimer tc(h, s, resident ? std::string("conv1_res_f32_kernel") : std::string(a.in2 ? "conv1_ring_f32_kernel<true, 256, 128, " : "conv1_ring_f32_kernel<false, 256, 128, ") + tname + ">", 2.0 * px * cin * pl, px * 4.0 * (cin + pl), 0.0);   // (as rocprofv3 prints them)
                            static unsigned attr_c1[2] = {0, 0};
                            const void* const fn = a.in2 ? reinterpret_cast<const void*>(conv1_ring_f32_kernel<true, 256, 128, T>) : reinterpret_cast<const void*>(conv1_ring_f32_kernel<false, 256, 128, T>);
                            if (first_use_on_this_device(attr_c1[a.in2 ? 1 : 0]))
                                DF3D_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_BYTES));
                            if (c.M % 128) {
                                df3d::set_error("conv1 of the split bottleneck needs whole 128-pixel tiles (M = %lld)", c.M);
                                return DF3D_EINVAL;
                            }
                            const unsigned c1_grid = (unsigned)std::min<long long>(c.M / 128, 2LL * cu_count());   // persistent: two workgroups per CU
                            if (resident) {
                                if constexpr (std::is_same<T, float>::value) {
                                    static unsigned attr_c1r = 0;
                                    if (first_use_on_this_device(attr_c1r))
                                        DF3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_res_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C1R_LDS_BYTES));
                                    c.wstream = reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream_w2d + WN_STREAM_BYTES;
                                    hipLaunchKernelGGL(conv1_res_f32_kernel, dim3((unsigned)std::min<long long>(c.M / 128, (long long)(cu_count() & ~7))), dim3(256), C1R_LDS_BYTES, s, c);
                                }
                            } else if (a.in2)
                                hipLaunchKernelGGL((conv1_ring_f32_kernel<true, 256, 128, T>), dim3(c1_grid), dim3(256), C1_LDS_BYTES, s, c);
                            else
                                hipLaunchKernelGGL((conv1_ring_f32_kernel<false, 256, 128, T>), dim3(c1_grid), dim3(256), C1_LDS_BYTES, s, c);
                            DF3D_LAUNCH_CHECK();
                        }
                    }
                    if (split && std::is_same<T, float>::value && r.w2d) {   // option `wino`: the tail with its 3x3 in the Winograd domain
                        const int blocks = n * (ti.h / BT_TH) * (ti.w / BT_TW);
                        // FLOPs: the direct form's (what the block computes, in the reference's terms); the kernel EXECUTES 16/36 of the 3x3's
                        ScopedTimer tm(h, s, std::string("bottleneck_wino_f32_kernel<") + (a.in2 ? "true, false, false>" : a.add2 ? "false, true, false>" : "false, false, false>"),
                                       2.0 * px * (9.0 * pl * pl + 2.0 * pl * pl), px * eb * (cin + 2.0 * pl + pl), st.m1_elems * n * eb, 2.0 * px * (4.0 * pl * pl + 2.0 * pl * pl));
                        const int rc = a.in2 ? launch_wino_f32<true, false>(r, blocks, s) : a.add2 ? launch_wino_f32<false, true>(r, blocks, s) : launch_wino_f32<false, false>(r, blocks, s);
                        if (rc) return rc;
                        break;
                    }
                    const char* const flags2 = split ? (a.in2 ? "true, false, true, " : a.add2 ? "false, true, true, " : "false, false, true, ")
                                                     : a.in2 ? "true, false, false, " : a.add2 ? "false, true, false, " : "false, false, false, ";
                    ScopedTimer tm(h, s, eb == 2 ? std::string("bottleneck_ring_kernel<") + tname + (a.in2 ? ", true, 256, false, " : a.add2 ? ", false, 256, true, " : ", false, 256, false, ") + std::to_string(ring_mode(r, h->ring2)) + ">"
                                                 : std::string("bottleneck_ring_f32_kernel<") + flags2 + tname + ">",   // as rocprofv3 prints them
                                   2.0 * px * ((split ? 0.0 : (double)cin * pl) + 9.0 * pl * pl + 2.0 * pl * pl), px * eb * (cin + 2.0 * pl + (split ? pl : 0)), st.m1_elems * n * eb);
                    const int blocks = n * (ti.h / BT_TH) * (ti.w / BT_TW);
                    int lds_bytes = BR_LDS_BYTES;
#ifdef DF3D_BT_TIMING
                    if (const char* e = getenv("BR_LDS")) lds_bytes = atoi(e);   // development: force one workgroup per CU (> 80 KB)
#endif
#ifdef BR_FORCE_LDS
                    lds_bytes = BR_FORCE_LDS;   // development builds: one workgroup per CU (> 80 KB)
#endif
                    int rc;
                    if constexpr (sizeof(T) == 2)
                        rc = a.in2 ? launch_ring_lp<T, true, 256>(r, h->ring2, blocks, lds_bytes, s) : a.add2 ? launch_ring_lp<T, false, 256, true>(r, h->ring2, blocks, lds_bytes, s)
                                                                                           : launch_ring_lp<T, false, 256>(r, h->ring2, blocks, lds_bytes, s);
                    else
                        rc = split ? (a.in2 ? launch_ring_f32<T, true, false, true>(r, blocks, lds_bytes, s)
                                      : a.add2 ? launch_ring_f32<T, false, true, true>(r, blocks, lds_bytes, s) : launch_ring_f32<T, false, false, true>(r, blocks, lds_bytes, s))
                             : a.in2 ? launch_ring_f32<T, true>(r, blocks, lds_bytes, s)
                             : a.add2 ? launch_ring_f32<T, false, true>(r, blocks, lds_bytes, s)
                                      : launch_ring_f32<T, false>(r, blocks, lds_bytes, s);
                    if (rc) return rc;
                    break;
                }
                ScopedTimer tm(h, s, std::string("bottleneck_kernel<") + tname + ", " + std::to_string(cin) + ", " + std::to_string(pl) + ", " + (ds ? "true" : "false") + ", " + (a.in2 ? "true" : "false") + ", " + (a.add2 ? "true" : "false") + ">",
                               2.0 * px * ((double)cin * pl + 9.0 * pl * pl + 2.0 * pl * pl + (ds ? 2.0 * cin * pl : 0.0)), px * eb * (cin + 2.0 * pl), st.m1_elems * n * eb);
                const int blocks = n * (ti.h / BT_TH) * (ti.w / BT_TW);
                if (int rc = launch_bottleneck<T>(a, cin, pl, blocks, s)) return rc;
                break;
            }
            case ST_HEAD: {
                const TensorDesc& ti = h->tensors[st.in];
                HeadArgs a;
                a.r = tptr(st.in);
                a.x = st.last ? nullptr : tptr(st.res);
                a.out = st.last ? nullptr : tptr(st.out);
                a.heat = st.last ? heatmaps : nullptr;
                a.wfc = wb + st.conv.w_off * eb;
                a.wsc = wb + st.conv2b.w_off * eb;
                a.bfc = h->blob + st.conv.b_off;
                a.bsc = h->blob + st.conv2b.b_off;
                a.wfc_ = st.last ? nullptr : wb + st.conv3b.w_off * eb;
                a.wsc_ = st.last ? nullptr : wb + st.conv4b.w_off * eb;
                a.bfc_ = st.last ? nullptr : h->blob + st.conv3b.b_off;
                a.bsc_ = st.last ? nullptr : h->blob + st.conv4b.b_off;
                a.fcstream = st.wstream >= 0 ? reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream : nullptr;
                a.fc2stream = st.wstream2 >= 0 && ((long long)n * ti.h * ti.w) % 128 == 0 ? reinterpret_cast<const unsigned char*>(h->lowp) + h->stream_base() + st.wstream2 : nullptr;
                a.M = (long long)n * ti.h * ti.w;
                a.HW = ti.h * ti.w;
                const void* fn = st.last ? reinterpret_cast<const void*>(head_kernel<T, true>) : reinterpret_cast<const void*>(head_kernel<T, false>);
                using HeadLast = HeadCfg<T, true>;
                using HeadMid = HeadCfg<T, false>;
                const int head_lds = st.last ? HeadLast::LDS_BYTES : HeadMid::LDS_BYTES;
                static unsigned attr_done[2] = {0, 0};
                if (first_use_on_this_device(attr_done[st.last]))
                    DF3D_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, head_lds));
                const double mm = (double)a.M;
                const double fl = 2.0 * mm * (256.0 * 256 + 256.0 * 19 + (st.last ? 0.0 : 256.0 * 256 + 19.0 * 256));
                ScopedTimer tm(h, s, std::string("head_kernel<") + tname + ", " + (st.last ? "true" : "false") + ">", fl,
                               mm * eb * (st.last ? 256.0 : 768.0) + (st.last ? mm * 19 * 4 : 0.0), st.m1_elems * n * eb);
                const unsigned blocks = (unsigned)((a.M + 127) / 128);
                if (st.last)
                    hipLaunchKernelGGL((head_kernel<T, true>), dim3(blocks), dim3(256), head_lds, s, a);
                else
                    hipLaunchKernelGGL((head_kernel<T, false>), dim3(blocks), dim3(256), head_lds, s, a);
                DF3D_LAUNCH_CHECK();
                break;
            }
            case ST_POOL: {
                const TensorDesc& to = h->tensors[st.out];
                const int chunks = to.pitch * eb / 16;
                const long long total = (long long)n * to.h * to.w * chunks;
                ScopedTimer tm(h, s, std::string("pool2_kernel<") + TypeName<StorageT<T>>::value + ">", 0.0, (double)total * 16 * 5, st.m1_elems * n * eb);
                hipLaunchKernelGGL((pool2_kernel<StorageT<T>>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   reinterpret_cast<const u32x4*>(tptr(st.in)), reinterpret_cast<u32x4*>(tptr(st.out)),
                                   total, to.h, to.w, chunks);
                DF3D_LAUNCH_CHECK();
                break;
            }
            case ST_UPADD: {
                const TensorDesc& to = h->tensors[st.out];
                const int chunks = to.pitch * eb / 16;
                const long long total = (long long)n * to.h * to.w * chunks;
                ScopedTimer tm(h, s, std::string("upadd_kernel<") + TypeName<StorageT<T>>::value + ">", 0.0, (double)total * 16 * 2.25, st.m1_elems * n * eb);
                hipLaunchKernelGGL((upadd_kernel<StorageT<T>>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   reinterpret_cast<const u32x4*>(tptr(st.in)), reinterpret_cast<const u32x4*>(tptr(st.res)),
                                   reinterpret_cast<u32x4*>(tptr(st.out)), total, to.h, to.w, chunks);
                DF3D_LAUNCH_CHECK();
                break;
            }
        }
        return DF3D_OK;
    };
    // Chains: runs of consecutive full-resolution steps are walked in chunks of `chain_views` views, so that what one step
    // writes is still in the 256 MB Infinity Cache when the next one reads it (a whole 896-view batch moves 3.8 GB per
    // tensor: nothing survives from one launch to the next).  The steps of a chain only read tensors of their own view range.
    for (int i = 0; i < upto;) {
        int j = i + 1;
        const int cv = h->chain_views;
        if (cv > 0 && cv < n_all && h->chain_end[i] > i + 1) {
            j = std::min(h->chain_end[i], upto);
            for (int v0 = 0; v0 < n_all; v0 += cv)
                for (int k = i; k < j; ++k)
                    if (int rc = launch(k, v0, std::min(cv, n_all - v0))) return rc;
        } else if (int rc = launch(i, 0, n_all)) {
            return rc;
        }
        i = j;
    }
    return DF3D_OK;
}

int run_steps_dtype(df3d_hg* h, const float* images, int n, int upto, float* heatmaps, unsigned char* act, hipStream_t s) {
    switch (h->dtype) {
        case DF3D_DTYPE_F32: return run_steps<float>(h, images, n, upto, heatmaps, act, s);
        case DF3D_DTYPE_F32S: return run_steps<F32S>(h, images, n, upto, heatmaps, act, s);
        case DF3D_DTYPE_F16: return run_steps<_Float16>(h, images, n, upto, heatmaps, act, s);
        default: return run_steps<__hip_bfloat16>(h, images, n, upto, heatmaps, act, s);
    }
}

int check_forward_args(df3d_hg* h, const void* images, int n, void* ws, size_t ws_bytes) {
    DF3D_CHECK_ARG(h != nullptr, "null handle");
    if (!h->blob) {
        df3d::set_error("df3d_hg_forward: weights not set (call df3d_hg_set_weights first)");
        return DF3D_ESTATE;
    }
    DF3D_CHECK_ARG(n > 0, "n must be positive");
    DF3D_CHECK_ARG(images && ws, "null pointer");
    DF3D_CHECK_ARG(ws_bytes >= df3d_hg_workspace_bytes(h, n), "workspace too small");
    DF3D_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
    return DF3D_OK;
}

}  // namespace

extern "C" {

#ifdef DF3D_BT_TIMING
// development build only: read and clear the per-phase cycle sums of bottleneck_ring_kernel
int df3d_dbg_ring_cycles(unsigned long long* out8) {
    // (twelve counters since round 4: out8 must hold 12 values)
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hgk::br_dbg), 96) != hipSuccess) return -1;
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(hgk::br_dbg), z, 96) == hipSuccess ? 0 : -1;
}
#endif

int df3d_hg_create(int dtype, int num_stacks, df3d_hg** out) {
    DF3D_CHECK_ARG(out != nullptr, "null out");
    DF3D_CHECK_ARG(dtype == DF3D_DTYPE_F32 || dtype == DF3D_DTYPE_BF16 || dtype == DF3D_DTYPE_F16 || dtype == DF3D_DTYPE_F32S, "dtype must be DF3D_DTYPE_F32, DF3D_DTYPE_BF16, DF3D_DTYPE_F16 or DF3D_DTYPE_F32S");
    DF3D_CHECK_ARG(num_stacks >= 1 && num_stacks <= 8, "num_stacks must be in [1, 8]");
    df3d_hg* h = new df3d_hg();
    h->dtype = dtype;
    h->num_stacks = num_stacks;
    h->fuse_upadd = 1;  // measured: +4-5 % frames/s in bf16, +2 % in fp32, bit-identical results
    h->build();
    *out = h;
    return DF3D_OK;
}

void df3d_hg_destroy(df3d_hg* h) {
    if (!h) return;
    for (auto& t : h->timed) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    delete h;
}

int df3d_hg_set_input(df3d_hg* h, int height, int width) {
    DF3D_CHECK_ARG(h != nullptr, "null handle");
    DF3D_CHECK_ARG(height > 0 && width > 0 && height % 64 == 0 && width % 64 == 0, "input height and width must be multiples of 64");
    h->H = height;
    h->W = width;
    const float* blob = h->blob;
    const void* lowp = h->lowp;
    h->build();  // parameter manifest does not depend on the spatial size
    h->blob = blob;
    h->lowp = lowp;
    return DF3D_OK;
}

int df3d_hg_set_option(df3d_hg* h, const char* key, int value) {
    DF3D_CHECK_ARG(h && key, "null argument");
    if (!strcmp(key, "fuse")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "fuse must be 0 or 1");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'fuse' before df3d_hg_set_weights (it changes the parameter manifest)");
        h->fuse = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "fuse_upadd")) {
        DF3D_CHECK_ARG(value >= 0 && value <= 2, "fuse_upadd must be 0, 1 or 2");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'fuse_upadd' before df3d_hg_set_weights (it changes the plan)");
        h->fuse_upadd = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "l1")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "l1 must be 0 or 1");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'l1' before df3d_hg_set_weights (it changes the plan and the low-precision buffer)");
        h->l1 = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "ring")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "ring must be 0 or 1");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'ring' before df3d_hg_set_weights (it changes the low-precision buffer)");
        h->ring = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "w2d")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "w2d must be 0 or 1");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'w2d' before df3d_hg_set_weights (it changes the weight streams)");
        h->w2d = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "ring2")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "ring2 must be 0 or 1");
        h->ring2 = value;
        return DF3D_OK;
    }
    if (!strcmp(key, "split1")) {
        DF3D_CHECK_ARG(value == 0 || value == 1 || (value >= 8 && value < 16), "split1 must be 0, 1 or 8 + a mask (1 identity blocks, 2 layer1, 4 layer2)");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'split1' before df3d_hg_set_weights (it changes the plan and the weight streams)");
        h->split1 = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "wino")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "wino must be 0 or 1");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'wino' before df3d_hg_set_weights (it changes the weight streams)");
        h->wino = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "c1res")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "c1res must be 0 or 1");
        h->c1res = value;
        return DF3D_OK;
    }
    if (!strcmp(key, "no_reuse")) {
        DF3D_CHECK_ARG(value == 0 || value == 1, "no_reuse must be 0 or 1");
        DF3D_CHECK_ARG(h->blob == nullptr, "set 'no_reuse' before df3d_hg_set_weights (it changes the workspace plan)");
        h->no_reuse = value;
        h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "chain_views")) {
        DF3D_CHECK_ARG(value >= 0, "chain_views must be >= 0");
        DF3D_CHECK_ARG(h->blob == nullptr || (value > 0) == (h->chain_views > 0), "switch 'chain_views' on or off before df3d_hg_set_weights (it changes the workspace plan)");
        const bool replan = (value > 0) != (h->chain_views > 0);
        h->chain_views = value;
        if (replan) h->build();
        return DF3D_OK;
    }
    if (!strcmp(key, "row_bytes")) {
        DF3D_CHECK_ARG(value == 0 || value == 64 || value == 128, "row_bytes must be 0, 64 or 128");
        h->rb_override = value;
        return DF3D_OK;
    }
    df3d::set_error("df3d_hg_set_option: unknown key %s", key);
    return DF3D_EINVAL;
}

int df3d_hg_num_params(const df3d_hg* h) { return h ? (int)h->params.size() : 0; }

int df3d_hg_param_desc(const df3d_hg* h, int i, df3d_hg_param* out) {
    DF3D_CHECK_ARG(h && out, "null argument");
    DF3D_CHECK_ARG(i >= 0 && i < (int)h->params.size(), "index out of range");
    *out = h->params[i];
    return DF3D_OK;
}

size_t df3d_hg_blob_floats(const df3d_hg* h) { return h ? h->blob_floats : 0; }

size_t df3d_hg_lowp_bytes(const df3d_hg* h) {
    if (!h) return 0;
    // bf16: bf16 copy of the blob + the pre-swizzled weight streams of the ring bottlenecks; f32: the weight streams only
    return h->stream_base() + h->stream_bytes;
}

namespace {
// max |w| over the blob as an integer (the bit pattern of |x| orders like the value; an infinity or a NaN is >= 0x7f800000)
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
}  // namespace

int df3d_hg_set_weights(df3d_hg* h, const float* blob_dev, void* lowp_dev, void* stream) {
    DF3D_CHECK_ARG(h && blob_dev, "null argument");
    const float* const blob_caller = blob_dev;
    DF3D_CHECK_ARG((reinterpret_cast<uintptr_t>(blob_dev) & 255) == 0, "blob must be 256-byte aligned");
    if ((h->dtype == DF3D_DTYPE_F16 || h->dtype == DF3D_DTYPE_F32S) && lowp_dev != nullptr) {
        // the half-precision engines need every operand inside the IEEE-half range: refuse weights (biases and folded BatchNorm vectors included)
        // that are not -- here, with the number, instead of as inf / NaN heat-maps later (one 4-byte read-back; the first word of the
        // caller's scratch buffer, which the packers below overwrite, is the reduction's cell)
        unsigned* const cell = reinterpret_cast<unsigned*>(lowp_dev);
        unsigned bits = 0;
        DF3D_HIP(hipMemsetAsync(cell, 0, 4, df3d::as_stream(stream)));
        hipLaunchKernelGGL(absmax_bits_kernel, dim3(256), dim3(256), 0, df3d::as_stream(stream), blob_dev, h->blob_floats, cell);
        DF3D_LAUNCH_CHECK();
        DF3D_HIP(hipMemcpyAsync(&bits, cell, 4, hipMemcpyDeviceToHost, df3d::as_stream(stream)));
        DF3D_HIP(hipStreamSynchronize(df3d::as_stream(stream)));
        float absmax;
        memcpy(&absmax, &bits, 4);
        if (!(absmax <= 65504.0f)) {   // (also true for an infinity or a NaN among the weights)
            df3d::set_error("max |w| = %g: the %s hourglass engine needs every weight inside the IEEE-half range (65504): use DF3D_DTYPE_F32 (or BF16)",
                            (double)absmax, h->dtype == DF3D_DTYPE_F16 ? "F16" : "F32S");
            return DF3D_EINVAL;
        }
    }
    if (h->lp()) {
        DF3D_CHECK_ARG(lowp_dev != nullptr, "a 16-bit engine needs a df3d_hg_lowp_bytes() device buffer");
        DF3D_CHECK_ARG((reinterpret_cast<uintptr_t>(lowp_dev) & 255) == 0, "lowp buffer must be 256-byte aligned");
        // the 16-bit copy of the blob; the 16-bit stem wants its weights as a [64][184] tile: overwrite the stem's slot of the copy
        if (h->dtype == DF3D_DTYPE_F16) {
            hipLaunchKernelGGL((f32_to_lp_kernel<_Float16>), dim3(1024), dim3(256), 0, df3d::as_stream(stream), blob_dev,
                               reinterpret_cast<unsigned short*>(lowp_dev), h->blob_floats);
            hipLaunchKernelGGL((stem_relayout_kernel<_Float16>), dim3((64 * 184 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                               blob_dev + h->steps[0].conv.w_off, reinterpret_cast<unsigned short*>(lowp_dev) + h->steps[0].conv.w_off);
        } else {
            hipLaunchKernelGGL((f32_to_lp_kernel<__hip_bfloat16>), dim3(1024), dim3(256), 0, df3d::as_stream(stream), blob_dev,
                               reinterpret_cast<unsigned short*>(lowp_dev), h->blob_floats);
            hipLaunchKernelGGL((stem_relayout_kernel<__hip_bfloat16>), dim3((64 * 184 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                               blob_dev + h->steps[0].conv.w_off, reinterpret_cast<unsigned short*>(lowp_dev) + h->steps[0].conv.w_off);
        }
        // weight streams of the ring bottlenecks: stage-by-stage LDS images (hg_bt_ring.h), from the 16-bit copy (byte movers:
        // the same kernels serve both formats)
        for (const Step& st : h->steps) {
            if (st.kind == ST_HEAD && st.wstream >= 0) {
                hipLaunchKernelGGL(bt_fc_pack_kernel, dim3((HD_FC_STAGES * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                                   reinterpret_cast<const unsigned short*>(lowp_dev) + st.conv.w_off,
                                   reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
                if (st.wstream2 >= 0)
                    hipLaunchKernelGGL(bt_fc2_pack_kernel, dim3((HD_FC2_STAGES * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                                       reinterpret_cast<const unsigned short*>(lowp_dev) + st.conv3b.w_off,
                                       reinterpret_cast<const unsigned short*>(lowp_dev) + st.conv4b.w_off,
                                       reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream2);
                continue;
            }
            if (st.kind != ST_BOTTLENECK || st.wstream < 0) continue;
            const unsigned short* lp = reinterpret_cast<const unsigned short*>(lowp_dev);
            if (st.l1) {
                hipLaunchKernelGGL(bt_l1_pack_kernel, dim3((L1_W_BYTES / 16 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                                   lp + st.conv.w_off, lp + st.conv2b.w_off, lp + st.conv3b.w_off, lp + st.conv4b.w_off,
                                   reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
                continue;
            }
            if (st.wstream_w2d >= 0)
                hipLaunchKernelGGL(bt_w2d_pack_kernel, dim3((BR_W2D_GROUPS * 256 + 255) / 256), dim3(256), 0, df3d::as_stream(stream), lp + st.conv2b.w_off,
                                   reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_w2d);
            const bool dsb = st.conv.cin != 2 * st.conv.cout;   // layer2: 128 -> 128 -> 128 -> 256 with the skip convolution
            hipLaunchKernelGGL(bt_ring_pack_kernel, dim3((br_nstage(st.conv.cin, dsb) * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                               lp + st.conv.w_off, lp + st.conv2b.w_off, lp + st.conv3b.w_off, dsb ? lp + st.conv4b.w_off : nullptr, st.conv.cin,
                               reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
        }
        DF3D_LAUNCH_CHECK();
        h->lowp = lowp_dev;
    } else if (h->dtype == DF3D_DTYPE_F32S && lowp_dev == nullptr) {
        df3d::set_error("an f32s engine needs a df3d_hg_lowp_bytes() device buffer (the pre-split copy of the weights)");
        return DF3D_EINVAL;
    } else if (h->stream_bytes && lowp_dev == nullptr) {
        // round 1's contract for f32 engines (no scratch buffer): honoured by falling back to the register-staged kernels, which
        // need no weight streams and give bit-identical results.  The parameter manifest does not depend on the option, so the
        // caller's blob stays valid.
        h->ring = 0;
        h->build();
    } else if (h->stream_bytes || h->dtype == DF3D_DTYPE_F32S) {
        DF3D_CHECK_ARG((reinterpret_cast<uintptr_t>(lowp_dev) & 255) == 0, "lowp buffer must be 256-byte aligned");
        if (h->dtype == DF3D_DTYPE_F32S) {
            // f32s: the weights pre-split per 16-float K step (hg_kernels.h f32s_presplit_kernel) -- a float32-sized copy of the blob in front of
            // the streams; the packers below then read THAT copy (they move whole 16-byte chunks and keep a chunk's index inside its step).
            // Biases and BatchNorm vectors are transformed along with the rest and never read from the copy.
            hipLaunchKernelGGL(f32s_presplit_kernel, dim3(1024), dim3(256), 0, df3d::as_stream(stream), reinterpret_cast<const u32x4*>(blob_dev),
                               reinterpret_cast<u32x4*>(lowp_dev), h->blob_floats / 16);
            // the stem's weights: two half-precision [64][184] tiles (hi, lo) in its slot of the copy (exactly the slot's 47 104 bytes)
            hipLaunchKernelGGL(stem_relayout_f32s_kernel, dim3((64 * 184 + 255) / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + h->steps[0].conv.w_off,
                               reinterpret_cast<unsigned short*>(reinterpret_cast<float*>(lowp_dev) + h->steps[0].conv.w_off));
            blob_dev = reinterpret_cast<const float*>(lowp_dev);   // (restored below: h->blob stays the caller's float32 blob)
        }
        for (const Step& st : h->steps) {
            if (st.kind == ST_HEAD && st.wstream >= 0)
                hipLaunchKernelGGL(bt_fc_pack_f32_kernel, dim3((HD_FC_STAGES_F32 * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                                   blob_dev + st.conv.w_off, reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
            if (st.kind != ST_BOTTLENECK || st.wstream < 0) continue;
            if (st.l2f) {
                hipLaunchKernelGGL(bt_l2f_pack_kernel, dim3((L2F_NSTAGE * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv2b.w_off,
                                   blob_dev + st.conv3b.w_off, blob_dev + st.conv4b.w_off, reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
                hipLaunchKernelGGL(bt_c1_pack_f32_kernel, dim3(((128 / 16) * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv.w_off,
                                   reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_c1, 128, 128);
                if (st.wstream_w2d >= 0 && h->dtype == DF3D_DTYPE_F32) {
                    unsigned char* const ws = reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_w2d;
                    hipLaunchKernelGGL(bt_wino_pack_kernel, dim3(128 * 128 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv2b.w_off, reinterpret_cast<float*>(ws));
                    hipLaunchKernelGGL(bt_wino_pack_w3_kernel, dim3(BRF_W3_STAGES * 512 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv3b.w_off, ws + WN_U_BYTES);
                    hipLaunchKernelGGL(bt_wino_pack_w3_kernel, dim3(BRF_W3_STAGES * 512 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv4b.w_off, ws + WN_U_BYTES + WN_W3_BYTES);
                }
                continue;
            }
            if (st.l1f) {
                hipLaunchKernelGGL(bt_l1f_pack_kernel, dim3((L1F_NSTAGE * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv2b.w_off,
                                   blob_dev + st.conv3b.w_off, blob_dev + st.conv4b.w_off, reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
                hipLaunchKernelGGL(bt_c1_pack_f32_kernel, dim3(((64 / 16) * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv.w_off,
                                   reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_c1, 64, 64);
                if (st.wstream_w2d >= 0 && h->dtype == DF3D_DTYPE_F32) {
                    unsigned char* const ws = reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_w2d;
                    hipLaunchKernelGGL(l1_wino_pack_u_kernel, dim3(64 * 64 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv2b.w_off, reinterpret_cast<float*>(ws));
                    hipLaunchKernelGGL(l1_wino_pack_w_kernel, dim3(4 * 512 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv3b.w_off, ws + L1W_U_BYTES);
                    hipLaunchKernelGGL(l1_wino_pack_w_kernel, dim3(4 * 512 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv4b.w_off, ws + L1W_U_BYTES + L1W_W_BYTES);
                }
                continue;
            }
            hipLaunchKernelGGL(bt_ring_pack_f32_kernel, dim3((BRF_NSTAGE * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                               blob_dev + st.conv.w_off, blob_dev + st.conv2b.w_off, blob_dev + st.conv3b.w_off,
                               reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream);
            if (st.wstream_c1 >= 0)
                hipLaunchKernelGGL(bt_c1_pack_f32_kernel, dim3((C1_NSTAGE * 512 + 255) / 256), dim3(256), 0, df3d::as_stream(stream),
                                   blob_dev + st.conv.w_off, reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_c1);
            if (st.wstream_w2d >= 0 && h->dtype == DF3D_DTYPE_F32)
            {
                unsigned char* const ws = reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + st.wstream_w2d;
                hipLaunchKernelGGL(bt_wino_pack_kernel, dim3(128 * 128 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv2b.w_off, reinterpret_cast<float*>(ws));
                hipLaunchKernelGGL(bt_wino_pack_w3_kernel, dim3(BRF_W3_STAGES * 512 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv3b.w_off, ws + WN_U_BYTES);
                hipLaunchKernelGGL(c1r_pack_kernel, dim3(C1_NSTAGE * 512 / 256), dim3(256), 0, df3d::as_stream(stream), blob_dev + st.conv.w_off, ws + WN_STREAM_BYTES);
            }
        }
        if (h->uses_zero_page)
            DF3D_HIP(hipMemsetAsync(reinterpret_cast<unsigned char*>(lowp_dev) + h->stream_base() + h->zero_off, 0, 256, df3d::as_stream(stream)));
        DF3D_LAUNCH_CHECK();
        h->lowp = lowp_dev;
    }
    h->blob = blob_caller;
    return DF3D_OK;
}

size_t df3d_hg_workspace_bytes(const df3d_hg* h, int n) {
    if (!h || n <= 0) return 0;
    return h->act_elems_per_view * (size_t)n * h->elem_bytes() + 256;
}

int df3d_hg_forward(df3d_hg* h, const float* images_dev, int n, float* heatmaps_dev, void* workspace_dev,
                    size_t workspace_bytes, void* stream) {
    if (int rc = check_forward_args(h, images_dev, n, workspace_dev, workspace_bytes)) return rc;
    DF3D_CHECK_ARG(heatmaps_dev != nullptr, "null heatmaps");
    unsigned char* act = reinterpret_cast<unsigned char*>(workspace_dev);
    return run_steps_dtype(h, images_dev, n, (int)h->steps.size(), heatmaps_dev, act, df3d::as_stream(stream));
}

int df3d_hg_forward_u8(df3d_hg* h, const unsigned char* frames_dev, const unsigned char* flip_dev, int n, int frame_h, int frame_w, int frame_c,
                       const float* mean3_host, const float* std3_host, int resize, float* heatmaps_dev, void* workspace_dev, size_t workspace_bytes,
                       void* stream) {
    if (int rc = check_forward_args(h, frames_dev, n, workspace_dev, workspace_bytes)) return rc;
    DF3D_CHECK_ARG(heatmaps_dev != nullptr && mean3_host && std3_host, "null pointer");
    DF3D_CHECK_ARG(frame_h > 0 && frame_w > 0 && (frame_c == 1 || frame_c == 3), "bad frame shape (C must be 1 or 3)");
    DF3D_CHECK_ARG(resize >= DF3D_RESIZE_BILINEAR && resize <= DF3D_RESIZE_AREA, "resize must be one of DF3D_RESIZE_*");
    hgk::StemU8 u;
    u.nm.resize = resize;
    u.frames = frames_dev;
    u.flip = flip_dev;
    u.FH = frame_h;
    u.FW = frame_w;
    u.FC = frame_c;
    for (int c = 0; c < 3; ++c) {
        DF3D_CHECK_ARG(std3_host[c] != 0.0f, "std must be non-zero");
        u.nm.mean[c] = mean3_host[c];
        u.nm.inv_std[c] = 1.0f / std3_host[c];
    }
    h->u8in = u;
    unsigned char* act = reinterpret_cast<unsigned char*>(workspace_dev);
    const int rc = run_steps_dtype(h, nullptr, n, (int)h->steps.size(), heatmaps_dev, act, df3d::as_stream(stream));
    h->u8in.frames = nullptr;
    return rc;
}

int df3d_hg_work(const df3d_hg* h, int n, double* flops, double* bytes) {
    DF3D_CHECK_ARG(h && flops && bytes, "null argument");
    *flops = h->flops_per_view * n;
    *bytes = h->elems_per_view * n * h->elem_bytes();
    return DF3D_OK;
}

int df3d_hg_profile(df3d_hg* h, int enable) {
    DF3D_CHECK_ARG(h != nullptr, "null handle");
    h->profiling = enable != 0;
    for (auto& t : h->timed) {
        h->event_pool.push_back(t.a);
        h->event_pool.push_back(t.b);
    }
    h->timed.clear();
    return DF3D_OK;
}

int df3d_hg_profile_count(const df3d_hg* h) { return h ? (int)h->kernel_names.size() : 0; }

int df3d_hg_profile_read(df3d_hg* h, int kernel_class, char* name_buf, int buflen, double* ms, double* flops, double* bytes, double* bytes_m1,
                         int* launches) {
    DF3D_CHECK_ARG(h && name_buf && buflen > 0 && ms && flops && bytes && bytes_m1 && launches, "null argument");
    DF3D_CHECK_ARG(kernel_class >= 0 && kernel_class < (int)h->kernel_names.size(), "kernel_class out of range");
    snprintf(name_buf, buflen, "%s", h->kernel_names[kernel_class].c_str());
    *ms = *flops = *bytes = *bytes_m1 = 0.0;
    *launches = 0;
    for (auto& t : h->timed) {
        if (t.cls != kernel_class) continue;
        DF3D_HIP(hipEventSynchronize(t.b));
        float e = 0.f;
        DF3D_HIP(hipEventElapsedTime(&e, t.a, t.b));
        *ms += e;
        *flops += t.flops;
        *bytes += t.bytes;
        *bytes_m1 += t.bytes_m1;
        *launches += 1;
    }
    return DF3D_OK;
}

int df3d_hg_profile_executed_flops(df3d_hg* h, int kernel_class, double* flops_executed) {
    DF3D_CHECK_ARG(h && flops_executed, "null argument");
    DF3D_CHECK_ARG(kernel_class >= 0 && kernel_class < (int)h->kernel_names.size(), "kernel_class out of range");
    *flops_executed = 0.0;
    for (auto& t : h->timed)
        if (t.cls == kernel_class) *flops_executed += t.flops_executed;
    return DF3D_OK;
}

int df3d_hg_num_steps(const df3d_hg* h) { return h ? (int)h->steps.size() : 0; }

double df3d_hg_step_m1_bytes(const df3d_hg* h, int step, int n) {
    if (!h || step < 0 || step >= (int)h->steps.size() || n <= 0) return 0.0;
    return h->steps[step].m1_elems * n * h->elem_bytes();
}

int df3d_hg_step_desc(const df3d_hg* h, int step, char* name_buf, int buflen, int* hwc) {
    DF3D_CHECK_ARG(h && name_buf && hwc && buflen > 0, "null argument");
    DF3D_CHECK_ARG(step >= 0 && step < (int)h->steps.size(), "step out of range");
    const Step& st = h->steps[step];
    snprintf(name_buf, buflen, "%s", st.name.c_str());
    if (st.out >= 0) {
        const TensorDesc& t = h->tensors[st.out];
        hwc[0] = t.h;
        hwc[1] = t.w;
        hwc[2] = t.c;
    } else {
        hwc[0] = h->H / 4;
        hwc[1] = h->W / 4;
        hwc[2] = h->classes;
    }
    return DF3D_OK;
}

int df3d_hg_forward_upto(df3d_hg* h, const float* images_dev, int n, int upto, float* out_dev, void* workspace_dev,
                         size_t workspace_bytes, void* stream) {
    if (int rc = check_forward_args(h, images_dev, n, workspace_dev, workspace_bytes)) return rc;
    DF3D_CHECK_ARG(upto >= 1 && upto <= (int)h->steps.size(), "upto out of range");
    DF3D_CHECK_ARG(out_dev != nullptr, "null output");
    const Step& st = h->steps[upto - 1];
    unsigned char* act = reinterpret_cast<unsigned char*>(workspace_dev);
    hipStream_t s = df3d::as_stream(stream);
    // a final NCHW step writes straight into out_dev (as heat-maps)
    if (int rc = run_steps_dtype(h, images_dev, n, upto, out_dev, act, s)) return rc;
    if (st.out < 0) return DF3D_OK;
    const TensorDesc& t = h->tensors[st.out];
    const long long pixels = (long long)n * t.h * t.w;
    const long long total = pixels * t.c;
    const void* src = act + t.off * (size_t)n * h->elem_bytes();
    if (!h->lp())
        hipLaunchKernelGGL((export_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, out_dev, pixels, t.c, t.pitch);
    else if (h->dtype == DF3D_DTYPE_F16)
        hipLaunchKernelGGL((export_kernel<_Float16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, out_dev, pixels, t.c, t.pitch);
    else
        hipLaunchKernelGGL((export_kernel<__hip_bfloat16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, out_dev, pixels, t.c, t.pitch);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

}  // extern "C"
