// libdig_cpu.so, second part: plain-C++ builds of the entry points either side of the pre-training step (SURVEY.md 8(f) rows N1 / N3 / N4:
// input transform, K/V-cached decode + beam step + metrics, the fine-tune step's sequence attention / losses / embeddings, the GRU attention
// head).  Same contract as dig_cpu.cpp: names, argument lists, error codes and storage conventions of include/dig_hip.h, straightforward
// loops over host memory, `stream` ignored, every call synchronous; integer / byte work bit for bit, floating point to the tolerance of the
// operator tests.  Test infrastructure for a GPU-less container -- nothing under dig_amd/ loads it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

typedef void* hipStream_t;
enum { DIG_OK = 0, DIG_ERR_ARG = -1, DIG_ERR_ALIGN = -2, DIG_ERR_LAUNCH = -3, DIG_ERR_UNSUPPORTED = -4 };
typedef uint16_t bf16_t;

struct dig_dropout_t {
  unsigned k0, k1, thr;
  float scale;
  unsigned pk0, pk1, pthr;
  float pscale;
  int rows_per_sample;
};

namespace {

inline float bf2f(bf16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline bf16_t f2bf(float f) {                       // round to nearest even, as v_cvt_pk_bf16_f32
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
inline unsigned drop_hash(unsigned k0, unsigned k1, unsigned a, unsigned b) {
  unsigned x = a ^ k0;
  x ^= x >> 16; x *= 0x7feb352du;
  x += k1 + b * 0x9e3779b9u;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
inline bool drop_keep(unsigned k0, unsigned k1, unsigned a, unsigned b, unsigned thr) { return drop_hash(k0, k1, a, b) >= thr; }
constexpr float NEG_INF = -std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------------------------------------------------------- input transform (N3)
constexpr int PRECISION_BITS = 32 - 8 - 2;

inline double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for every output index of one axis (csrc/input.hip coeffs_for, same operation order)
void axis_coeffs(int in_size, int out_size, int ksize, std::vector<int>& kk, std::vector<int>& bounds) {
  kk.assign((size_t)out_size * ksize, 0);
  bounds.assign((size_t)out_size * 2, 0);
  for (int xx = 0; xx < out_size; ++xx) {
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const double center = ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter(((double)(x + xmin) - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
      double w = bicubic_filter(((double)(x + xmin) - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
      kk[(size_t)xx * ksize + x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}

inline int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

int ksize_for(int in_size, int out_size) {
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)std::ceil(2.0 * fs) * 2 + 1;
}

inline unsigned philox_first(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

inline long long clamp_tok(long long t, int vocab) { return t < 0 ? 0 : (t >= vocab ? vocab - 1 : t); }

// log-sum-exp of a row (max-shifted)
inline void row_max_sum(const float* x, int C, float& m, float& s) {
  m = NEG_INF;
  for (int c = 0; c < C; ++c) m = std::max(m, x[c]);
  s = 0.f;
  for (int c = 0; c < C; ++c) s += std::exp(x[c] - m);
}

inline unsigned char canon_of(long long v, const unsigned char* canon, int n_classes) { return (v >= 0 && v < n_classes) ? canon[v] : 0; }

}  // namespace

extern "C" {

int dig_resize_bicubic_normalize_u8(const unsigned char* packed, const long long* offsets, const int* heights, const int* widths, int n_img,
                                    float* out, int out_h, int out_w, float mean, float std_, int max_h, int max_w, hipStream_t) {
  if (!packed || !offsets || !heights || !widths || !out || n_img <= 0 || out_h <= 0 || out_w <= 0 || max_h <= 0 || max_w <= 0 || std_ == 0.f)
    return DIG_ERR_ARG;
  const int ksh = ksize_for(max_w, out_w), ksv = ksize_for(max_h, out_h);
  if (((size_t)out_w * (ksh + 2) + (size_t)out_h * (ksv + 2)) * sizeof(int) > 160 * 1024) return DIG_ERR_UNSUPPORTED;
#pragma omp parallel for
  for (int img = 0; img < n_img; ++img) {
    const int h = heights[img], w = widths[img];
    const unsigned char* src = packed + offsets[img];
    std::vector<int> kh, bh, kv, bv;
    axis_coeffs(w, out_w, ksh, kh, bh);
    axis_coeffs(h, out_h, ksv, kv, bv);
    const bool pass_h = (w != out_w), pass_v = (h != out_h);
    const size_t plane = (size_t)out_h * out_w;
    float* o = out + (size_t)img * 3 * plane;
    for (int p = 0; p < out_h * out_w; ++p) {
      const int yy = p / out_w, xx = p - yy * out_w;
      const int x0 = pass_h ? bh[2 * xx] : xx, nx = pass_h ? bh[2 * xx + 1] : 1;
      const int y0 = pass_v ? bv[2 * yy] : yy, ny = pass_v ? bv[2 * yy + 1] : 1;
      const int* kx = kh.data() + (size_t)xx * ksh;
      const int* ky = kv.data() + (size_t)yy * ksv;
      int a[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
      int r[3] = {0, 0, 0};
      for (int y = 0; y < ny; ++y) {
        const unsigned char* row = src + ((size_t)(y0 + y) * w + x0) * 3;
        int hv[3];
        if (pass_h) {
          int s[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
          for (int x = 0; x < nx; ++x)
            for (int c = 0; c < 3; ++c) s[c] += (int)row[3 * x + c] * kx[x];
          for (int c = 0; c < 3; ++c) hv[c] = clip8(s[c]);
        } else {
          for (int c = 0; c < 3; ++c) hv[c] = row[c];
        }
        for (int c = 0; c < 3; ++c) {
          if (pass_v) a[c] += hv[c] * ky[y];
          else r[c] = hv[c];
        }
      }
      for (int c = 0; c < 3; ++c) {
        if (pass_v) r[c] = clip8(a[c]);
        o[c * plane + p] = ((float)r[c] / 255.0f - mean) / std_;
      }
    }
  }
  return DIG_OK;
}

int dig_random_masks(unsigned char* mask, int n_rows, int n_patches, int num_mask, unsigned long long seed, unsigned step, hipStream_t) {
  if (!mask || n_rows <= 0 || n_patches <= 0 || n_patches > 16384 || num_mask < 0 || num_mask > n_patches) return DIG_ERR_ARG;
  const unsigned k0 = (unsigned)(seed & 0xffffffffull), k1 = (unsigned)(seed >> 32);
#pragma omp parallel for
  for (int r = 0; r < n_rows; ++r) {
    std::vector<unsigned> keys(n_patches);
    for (int p = 0; p < n_patches; ++p) keys[p] = philox_first((unsigned)r, (unsigned)p, step, 0u, k0, k1);
    for (int p = 0; p < n_patches; ++p) {
      int rank = 0;
      for (int q = 0; q < n_patches; ++q) rank += (keys[q] < keys[p]) || (keys[q] == keys[p] && q < p);
      mask[(size_t)r * n_patches + p] = rank < num_mask ? 1 : 0;
    }
  }
  return DIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- decode (N4)
int dig_decode_embed(const long long* tokens, const float* emb, const float* pe_row, void* x_, int B, int d, int vocab, hipStream_t) {
  if (!tokens || !emb || !pe_row || !x_ || B <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  bf16_t* x = (bf16_t*)x_;
  for (int b = 0; b < B; ++b) {
    const long long t = clamp_tok(tokens[b], vocab);
    for (int c = 0; c < d; ++c) x[(size_t)b * d + c] = f2bf(emb[(size_t)t * d + c] + pe_row[c]);
  }
  return DIG_OK;
}

int dig_decode_self_attn(const void* qkv_cache, void* out_, int B, int T, int heads, int head_dim, int t, float scale, hipStream_t) {
  if (!qkv_cache || !out_ || B <= 0 || heads <= 0 || t < 0 || t >= T) return DIG_ERR_ARG;
  if (head_dim != 64) return DIG_ERR_UNSUPPORTED;
  const bf16_t* qkv = (const bf16_t*)qkv_cache;
  bf16_t* out = (bf16_t*)out_;
  const int hk = heads * 64;
  const size_t row = (size_t)3 * hk;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const bf16_t* base = qkv + (size_t)b * T * row + h * 64;
      std::vector<float> s(t + 1);
      float m = NEG_INF;
      for (int j = 0; j <= t; ++j) {
        float a = 0.f;
        for (int c = 0; c < 64; ++c) a += bf2f(base[(size_t)t * row + c]) * scale * bf2f(base[(size_t)j * row + hk + c]);
        s[j] = a;
        m = std::max(m, a);
      }
      float l = 0.f;
      for (int j = 0; j <= t; ++j) { s[j] = std::exp(s[j] - m); l += s[j]; }
      for (int c = 0; c < 64; ++c) {
        float a = 0.f;
        for (int j = 0; j <= t; ++j) a += s[j] * bf2f(base[(size_t)j * row + 2 * hk + c]);
        out[(size_t)b * hk + h * 64 + c] = f2bf(a / l);
      }
    }
  return DIG_OK;
}

int dig_decode_cross_attn(const void* q_, const void* kv_mem, void* out_, float* weights, int B, int n_mem, int heads, int head_dim, float scale,
                          int slots_per_mem, hipStream_t) {
  if (!q_ || !kv_mem || !out_ || B <= 0 || n_mem <= 0 || heads <= 0 || slots_per_mem < 1 || B % slots_per_mem) return DIG_ERR_ARG;
  if (head_dim != 64 || n_mem > 8192) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(kv_mem)) return DIG_ERR_ALIGN;
  const bf16_t* q = (const bf16_t*)q_;
  const bf16_t* kv = (const bf16_t*)kv_mem;
  bf16_t* out = (bf16_t*)out_;
  const int hk = heads * 64;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const bf16_t* kb = kv + (size_t)(b / slots_per_mem) * n_mem * 2 * hk + h * 64;
      std::vector<float> s(n_mem);
      float m = NEG_INF;
      for (int j = 0; j < n_mem; ++j) {
        float a = 0.f;
        for (int c = 0; c < 64; ++c) a += bf2f(q[(size_t)b * hk + h * 64 + c]) * scale * bf2f(kb[(size_t)j * 2 * hk + c]);
        s[j] = a;
        m = std::max(m, a);
      }
      float l = 0.f;
      for (int j = 0; j < n_mem; ++j) { s[j] = std::exp(s[j] - m); l += s[j]; }
      const float inv = 1.f / l;
      if (weights)
        for (int j = 0; j < n_mem; ++j) weights[((size_t)b * heads + h) * n_mem + j] = s[j] * inv;
      for (int c = 0; c < 64; ++c) {
        float a = 0.f;
        for (int j = 0; j < n_mem; ++j) a += s[j] * bf2f(kb[(size_t)j * 2 * hk + hk + c]);
        out[(size_t)b * hk + h * 64 + c] = f2bf(a * inv);
      }
    }
  return DIG_OK;
}

int dig_softmax_argmax(const float* logits, int ld, float* probs, long long* tokens, int B, int C, hipStream_t) {
  if (!logits || !probs || !tokens || B <= 0 || C <= 0 || ld < C) return DIG_ERR_ARG;
  for (int b = 0; b < B; ++b) {
    const float* row = logits + (size_t)b * ld;
    float m = NEG_INF;
    int am = 0x7fffffff;
    for (int c = 0; c < C; ++c)
      if (row[c] > m) { m = row[c]; am = c; }                 // first index of the maximum (torch.max)
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += std::exp(row[c] - m);
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) probs[(size_t)b * C + c] = std::exp(row[c] - m) * inv;
    tokens[b] = am;
  }
  return DIG_OK;
}

int dig_beam_step(const float* logits, int ld, float* seq_scores, int B, int beam_width, int C, int eos, long long* symbols,
                  long long* predecessors, float* stored_scores, hipStream_t) {
  if (!logits || !seq_scores || !symbols || !predecessors || !stored_scores || B <= 0 || C <= 0 || ld < C) return DIG_ERR_ARG;
  if (beam_width < 1 || beam_width > 16 || (size_t)beam_width * C * sizeof(float) > 60000) return DIG_ERR_UNSUPPORTED;
  const int bw = beam_width, n = bw * C;
  for (int b = 0; b < B; ++b) {
    std::vector<float> cand(n);
    for (int k = 0; k < bw; ++k) {
      const float* row = logits + (size_t)(b * bw + k) * ld;
      float m, s;
      row_max_sum(row, C, m, s);
      const float lse = m + std::log(s);
      for (int c = 0; c < C; ++c) cand[k * C + c] = seq_scores[b * bw + k] + (row[c] - lse);
    }
    for (int r = 0; r < bw; ++r) {
      float bv = NEG_INF;
      int bi = 0x7fffffff;
      for (int i = 0; i < n; ++i)
        if (cand[i] > bv || (cand[i] == bv && i < bi)) { bv = cand[i]; bi = i; }
      if (bi == 0x7fffffff) bi = r;
      const int sym = bi % C;
      symbols[b * bw + r] = sym;
      predecessors[b * bw + r] = bi / C + b * bw;
      stored_scores[b * bw + r] = bv;
      seq_scores[b * bw + r] = sym == eos ? NEG_INF : bv;
      cand[bi] = NEG_INF;
    }
  }
  return DIG_OK;
}

int dig_string_match(const long long* pred, const long long* target, const unsigned char* canon, int n_classes, int eos, int B, int T,
                     unsigned char* match, hipStream_t) {
  if (!pred || !target || !canon || !match || n_classes <= 0 || B <= 0 || T <= 0) return DIG_ERR_ARG;
  for (int b = 0; b < B; ++b) {
    // the kept characters of both rows (cut at eos, dropped classes removed), compared as strings
    std::vector<unsigned char> ps, ts;
    for (int i = 0; i < T && pred[(size_t)b * T + i] != eos; ++i) {
      const unsigned char c = canon_of(pred[(size_t)b * T + i], canon, n_classes);
      if (c) ps.push_back(c);
    }
    for (int i = 0; i < T && target[(size_t)b * T + i] != eos; ++i) {
      const unsigned char c = canon_of(target[(size_t)b * T + i], canon, n_classes);
      if (c) ts.push_back(c);
    }
    match[b] = ps == ts ? 1 : 0;
  }
  return DIG_OK;
}

int dig_char_fmeasure(const long long* pred, const long long* target, const unsigned char* canon, int n_classes, int eos, int B, int T,
                      double* f_per_sample, hipStream_t) {
  if (!pred || !target || !canon || !f_per_sample || n_classes <= 0 || B <= 0 || T <= 0) return DIG_ERR_ARG;
  for (int b = 0; b < B; ++b) {
    unsigned long long ps = 0, ts = 0;
    for (int i = 0; i < T; ++i) {
      const long long v = pred[(size_t)b * T + i];
      if (v == eos) break;
      const unsigned char c = canon_of(v, canon, n_classes);
      if (c) ps |= 1ull << (c & 63);
    }
    for (int i = 0; i < T; ++i) {
      const long long v = target[(size_t)b * T + i];
      if (v == eos) break;
      const unsigned char c = canon_of(v, canon, n_classes);
      if (c) ts |= 1ull << (c & 63);
    }
    const double n = (double)__builtin_popcountll(ps & ts);
    const double p = n / ((double)__builtin_popcountll(ps) + 1e-5), r = n / ((double)__builtin_popcountll(ts) + 1e-5);
    f_per_sample[b] = 2 * p * r / (p + r + 1e-5);
  }
  return DIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- sequence losses
int dig_seq_cross_entropy(const float* input, const long long* target, const long long* length, int B, int T, int C, float* row_workspace,
                          float* loss, hipStream_t) {
  if (!input || !target || !length || !row_workspace || !loss || B <= 0 || T <= 0 || C <= 0) return DIG_ERR_ARG;
  double total = 0.0;
  for (int row = 0; row < B * T; ++row) {
    const int b = row / T, t = row - b * T;
    float v = 0.f;
    if (t < length[b]) {
      const float* x = input + (size_t)row * C;
      float m, s;
      row_max_sum(x, C, m, s);
      const long long y = clamp_tok(target[row], C);
      v = -(x[y] - m - std::log(s));
    }
    row_workspace[row] = v;
    total += v;
  }
  loss[0] = (float)total * (1.0f / (float)B);
  return DIG_OK;
}

int dig_seq_ls_cross_entropy(const float* input, const long long* target, const long long* length, int B, int T, int C, float smoothing,
                             float* row_workspace, float* loss, hipStream_t) {
  if (!input || !target || !length || !row_workspace || !loss || B <= 0 || T <= 0 || C <= 0 || smoothing < 0.f || smoothing > 1.f) return DIG_ERR_ARG;
  const int n = B * T;
  double a = 0.0, sj = 0.0, cnt = 0.0;
  for (int row = 0; row < n; ++row) {
    const int b = row / T, t = row - b * T;
    const float* x = input + (size_t)row * C;
    float m, s, sx = 0.f;
    row_max_sum(x, C, m, s);
    for (int c = 0; c < C; ++c) sx += x[c];
    const float lse = m + std::log(s);
    const long long y = clamp_tok(target[row], C);
    row_workspace[row] = t < length[b] ? lse - x[y] : 0.f;
    row_workspace[n + row] = lse - sx / (float)C;
    a += row_workspace[row];
    sj += row_workspace[n + row];
  }
  for (int b = 0; b < B; ++b) { const long long l = length[b]; cnt += (double)(l < 0 ? 0 : (l > T ? T : l)); }
  loss[0] = (float)(((double)(1.f - smoothing) * (double)n * a + (double)smoothing * cnt * sj) / (double)B);
  return DIG_OK;
}

int dig_seq_ls_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar, int B,
                                 int T, int C, float smoothing, void* dlogits_, int ldd, hipStream_t) {
  if (!logits || !target || !length || !dlogits_ || B <= 0 || T <= 0 || C <= 0 || ld < C || ldd < C || smoothing < 0.f || smoothing > 1.f) return DIG_ERR_ARG;
  bf16_t* dlogits = (bf16_t*)dlogits_;
  float cnt = 0.f;
  for (int b = 0; b < B; ++b) { const long long l = length[b]; cnt += (float)(l < 0 ? 0 : (l > T ? T : l)); }
  const float confidence = 1.f - smoothing, sc = (gscalar ? gscalar[0] : 1.f) / (float)B, ws = smoothing * cnt;
  for (int row = 0; row < B * T; ++row) {
    const int b = row / T, t = row - b * T;
    const float* x = logits + (size_t)row * ld;
    float m, s;
    row_max_sum(x, C, m, s);
    const float inv = 1.f / s, wn = t < length[b] ? confidence * (float)(B * T) : 0.f;
    const long long y = clamp_tok(target[row], C);
    for (int c = 0; c < ldd; ++c) {
      float v = 0.f;
      if (c < C) {
        const float p = std::exp(x[c] - m) * inv;
        v = sc * (wn * (p - (c == y ? 1.f : 0.f)) + ws * (p - 1.f / (float)C));
      }
      dlogits[(size_t)row * ldd + c] = f2bf(v);
    }
  }
  return DIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- fine-tune step (N1): sequence attention
// per (sample, head): logits = (q . k) * scale, masked keys -> probability 0, lse of the unmasked logits kept; attention dropout multiplies
// the normalised probabilities (csrc/seq_attn.hip)
int dig_seq_attn_fwd_dropout(const void* q_, int ldq, const void* k_, int ldk, const void* v_, int ldv, void* out_, int ldo, float* lse, int B,
                             int heads, int Lq, int Lk, float scale, int causal, const long long* lens, const dig_dropout_t* drop, hipStream_t) {
  if (!q_ || !k_ || !v_ || !out_ || !lse || B <= 0 || heads <= 0 || Lq <= 0 || Lq > 32 || Lk <= 0 || Lk > 512) return DIG_ERR_ARG;
  if ((ldk & 7) || (ldv & 7) || !aligned16(k_) || !aligned16(v_)) return DIG_ERR_ALIGN;
  const bf16_t* q = (const bf16_t*)q_; const bf16_t* k = (const bf16_t*)k_; const bf16_t* v = (const bf16_t*)v_;
  bf16_t* out = (bf16_t*)out_;
  const bool dropping = drop && drop->thr;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const long long len = lens ? lens[b] : (long long)Lk;
      std::vector<float> S((size_t)Lk);
      for (int i = 0; i < Lq; ++i) {
        const bf16_t* qr = q + ((size_t)b * Lq + i) * ldq + h * 64;
        float m = NEG_INF;
        for (int j = 0; j < Lk; ++j) {
          const bool ok = (!causal || j <= i) && (!lens || j < len);
          float a = NEG_INF;
          if (ok) {
            a = 0.f;
            const bf16_t* kr = k + ((size_t)b * Lk + j) * ldk + h * 64;
            for (int c = 0; c < 64; ++c) a += bf2f(qr[c]) * scale * bf2f(kr[c]);
          }
          S[j] = a;
          m = std::max(m, a);
        }
        float sum = 0.f;
        for (int j = 0; j < Lk; ++j) {
          const float e = (S[j] == NEG_INF) ? 0.f : std::exp(S[j] - m);
          sum += e;
          S[j] = dropping ? (drop_keep(drop->k0, drop->k1, ((unsigned)i << 16) | (unsigned)j, (unsigned)(b * heads + h), drop->thr) ? e * drop->scale : 0.f) : e;
        }
        const float inv = 1.f / sum;
        lse[((size_t)b * heads + h) * Lq + i] = m + std::log(sum);
        for (int c = 0; c < 64; ++c) {
          float a = 0.f;
          for (int j = 0; j < Lk; ++j) a += S[j] * bf2f(v[((size_t)b * Lk + j) * ldv + h * 64 + c]);
          out[((size_t)b * Lq + i) * ldo + h * 64 + c] = f2bf(a * inv);
        }
      }
    }
  return DIG_OK;
}

int dig_seq_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, float* lse, int B, int heads,
                     int Lq, int Lk, float scale, int causal, const long long* lens, hipStream_t st) {
  return dig_seq_attn_fwd_dropout(q, ldq, k, ldk, v, ldv, out, ldo, lse, B, heads, Lq, Lk, scale, causal, lens, nullptr, st);
}

int dig_seq_attn_bwd_dropout(const void* q_, int ldq, const void* k_, int ldk, const void* v_, int ldv, const void* dout_, int ldo, const float* lse,
                             void* dq_, int lddq, void* dk_, int lddk, void* dv_, int lddv, int B, int heads, int Lq, int Lk, float scale, int causal,
                             const long long* lens, const dig_dropout_t* drop, hipStream_t) {
  if (!q_ || !k_ || !v_ || !dout_ || !lse || !dq_ || !dk_ || !dv_ || B <= 0 || heads <= 0 || Lq <= 0 || Lq > 32 || Lk <= 0 || Lk > 512) return DIG_ERR_ARG;
  if ((ldk & 7) || (ldv & 7) || !aligned16(k_) || !aligned16(v_)) return DIG_ERR_ALIGN;
  const bf16_t* q = (const bf16_t*)q_; const bf16_t* k = (const bf16_t*)k_; const bf16_t* v = (const bf16_t*)v_;
  const bf16_t* dout = (const bf16_t*)dout_;
  bf16_t* dq = (bf16_t*)dq_; bf16_t* dk = (bf16_t*)dk_; bf16_t* dv = (bf16_t*)dv_;
  const bool dropping = drop && drop->thr;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const long long len = lens ? lens[b] : (long long)Lk;
      std::vector<float> P((size_t)Lq * Lk), dS((size_t)Lq * Lk), F((size_t)Lq * Lk);
      for (int i = 0; i < Lq; ++i) {
        const bf16_t* qr = q + ((size_t)b * Lq + i) * ldq + h * 64;
        const bf16_t* gr = dout + ((size_t)b * Lq + i) * ldo + h * 64;
        const float l = lse[((size_t)b * heads + h) * Lq + i];
        float del = 0.f;
        for (int j = 0; j < Lk; ++j) {
          const bool ok = (!causal || j <= i) && (!lens || j < len);
          float pr = 0.f, dp = 0.f, f = 1.f;
          if (dropping) f = drop_keep(drop->k0, drop->k1, ((unsigned)i << 16) | (unsigned)j, (unsigned)(b * heads + h), drop->thr) ? drop->scale : 0.f;
          if (ok) {
            float sc = 0.f;
            const bf16_t* kr = k + ((size_t)b * Lk + j) * ldk + h * 64;
            const bf16_t* vr = v + ((size_t)b * Lk + j) * ldv + h * 64;
            for (int c = 0; c < 64; ++c) { sc += bf2f(qr[c]) * bf2f(kr[c]); dp += bf2f(gr[c]) * bf2f(vr[c]); }
            pr = std::exp(sc * scale - l);
            dp *= f;
          }
          P[(size_t)i * Lk + j] = pr; dS[(size_t)i * Lk + j] = dp; F[(size_t)i * Lk + j] = f;
          del += pr * dp;
        }
        for (int j = 0; j < Lk; ++j) dS[(size_t)i * Lk + j] = P[(size_t)i * Lk + j] * (dS[(size_t)i * Lk + j] - del);
      }
      for (int j = 0; j < Lk; ++j)
        for (int c = 0; c < 64; ++c) {
          float ak = 0.f, av = 0.f;
          for (int i = 0; i < Lq; ++i) {
            ak += dS[(size_t)i * Lk + j] * bf2f(q[((size_t)b * Lq + i) * ldq + h * 64 + c]);
            av += P[(size_t)i * Lk + j] * F[(size_t)i * Lk + j] * bf2f(dout[((size_t)b * Lq + i) * ldo + h * 64 + c]);
          }
          dk[((size_t)b * Lk + j) * lddk + h * 64 + c] = f2bf(ak * scale);
          dv[((size_t)b * Lk + j) * lddv + h * 64 + c] = f2bf(av);
        }
      for (int i = 0; i < Lq; ++i)
        for (int c = 0; c < 64; ++c) {
          float a = 0.f;
          for (int j = 0; j < Lk; ++j) a += dS[(size_t)i * Lk + j] * bf2f(k[((size_t)b * Lk + j) * ldk + h * 64 + c]);
          dq[((size_t)b * Lq + i) * lddq + h * 64 + c] = f2bf(a * scale);
        }
    }
  return DIG_OK;
}

int dig_seq_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo, const float* lse, void* dq,
                     int lddq, void* dk, int lddk, void* dv, int lddv, int B, int heads, int Lq, int Lk, float scale, int causal,
                     const long long* lens, hipStream_t st) {
  return dig_seq_attn_bwd_dropout(q, ldq, k, ldk, v, ldv, dout, ldo, lse, dq, lddq, dk, lddk, dv, lddv, B, heads, Lq, Lk, scale, causal, lens, nullptr, st);
}

int dig_seq_embed_fwd(const long long* tokens, const float* emb, const float* pos_table, void* x_, int B, int T, int d, int vocab, hipStream_t) {
  if (!tokens || !emb || !pos_table || !x_ || B <= 0 || T <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  bf16_t* x = (bf16_t*)x_;
  for (int r = 0; r < B * T; ++r) {
    const long long t = clamp_tok(tokens[r], vocab);
    for (int c = 0; c < d; ++c) x[(size_t)r * d + c] = f2bf(emb[(size_t)t * d + c] + pos_table[(size_t)(r % T) * d + c]);
  }
  return DIG_OK;
}

int dig_seq_embed_bwd_lens(const long long* tokens, const void* dx_, float* demb, int n_tok, int d, int vocab, int T, const long long* lens,
                           hipStream_t) {
  if (!tokens || !dx_ || !demb || n_tok <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  if (lens && (T <= 0 || n_tok % T)) return DIG_ERR_ARG;
  if ((size_t)n_tok * sizeof(int) > 60 * 1024) return DIG_ERR_UNSUPPORTED;
  const bf16_t* dx = (const bf16_t*)dx_;
  for (int v = 0; v < vocab; ++v) {
    std::vector<int> list;                                             // matching rows, ascending: a fixed summation order
    for (int r = 0; r < n_tok; ++r)
      if (tokens[r] == v && (!lens || (r % T) < lens[r / T])) list.push_back(r);
    const int n = (int)list.size();
    if (!n) continue;
    for (int c = 0; c < d; ++c) {
      float a0 = 0.f, a1 = 0.f;
      int k = 0;
      for (; k + 1 < n; k += 2) { a0 += bf2f(dx[(size_t)list[k] * d + c]); a1 += bf2f(dx[(size_t)list[k + 1] * d + c]); }
      if (k < n) a0 += bf2f(dx[(size_t)list[k] * d + c]);
      demb[(size_t)v * d + c] += a0 + a1;
    }
  }
  return DIG_OK;
}

int dig_seq_embed_bwd(const long long* tokens, const void* dx, float* demb, int n_tok, int d, int vocab, hipStream_t st) {
  return dig_seq_embed_bwd_lens(tokens, dx, demb, n_tok, d, vocab, 0, nullptr, st);
}

int dig_seq_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar, int B, int T,
                              int C, void* dlogits_, int ldd, hipStream_t) {
  if (!logits || !target || !length || !dlogits_ || B <= 0 || T <= 0 || C <= 0 || ld < C || ldd < C) return DIG_ERR_ARG;
  bf16_t* dlogits = (bf16_t*)dlogits_;
  const float sc = (gscalar ? gscalar[0] : 1.f) * (1.0f / (float)B);
  for (int row = 0; row < B * T; ++row) {
    const int b = row / T, t = row - b * T;
    bf16_t* out = dlogits + (size_t)row * ldd;
    if (t >= length[b]) { for (int c = 0; c < ldd; ++c) out[c] = 0; continue; }
    const float* x = logits + (size_t)row * ld;
    float m, s;
    row_max_sum(x, C, m, s);
    const float inv = 1.f / s;
    const long long y = clamp_tok(target[row], C);
    for (int c = 0; c < ldd; ++c) out[c] = c < C ? f2bf(sc * (std::exp(x[c] - m) * inv - (c == y ? 1.f : 0.f))) : (bf16_t)0;
  }
  return DIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- GRU attention head (N1)
// (csrc/gru_attn.hip; tanh / sigmoid in libm here, the device uses exp-based forms: fp32 round-off class)
int dig_addattn_fwd(const void* xproj_, const void* sproj_, const float* w, const void* x_, float* alpha, void* ctx_, int ldc, int B, int N, int A,
                    int X, hipStream_t) {
  if (!xproj_ || !sproj_ || !w || !x_ || !alpha || !ctx_ || B <= 0 || N <= 0 || N > 512 || A <= 0 || A > 1024 || (A & 7) || X <= 0 || (X & 1) || (ldc & 1))
    return DIG_ERR_ARG;
  if (!aligned16(xproj_)) return DIG_ERR_ALIGN;
  const bf16_t* xproj = (const bf16_t*)xproj_; const bf16_t* sproj = (const bf16_t*)sproj_; const bf16_t* x = (const bf16_t*)x_;
  bf16_t* ctx = (bf16_t*)ctx_;
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    std::vector<float> v(N);
    float m = NEG_INF;
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      for (int a = 0; a < A; ++a) acc += w[a] * std::tanh(bf2f(sproj[(size_t)b * A + a]) + bf2f(xproj[((size_t)b * N + n) * A + a]));
      v[n] = acc;
      m = std::max(m, acc);
    }
    float sum = 0.f;
    for (int n = 0; n < N; ++n) { v[n] = std::exp(v[n] - m); sum += v[n]; }
    const float inv = 1.f / sum;
    for (int n = 0; n < N; ++n) { v[n] *= inv; alpha[(size_t)b * N + n] = v[n]; }
    for (int c = 0; c < X; ++c) {
      float a = 0.f;
      for (int n = 0; n < N; ++n) a += v[n] * bf2f(x[((size_t)b * N + n) * X + c]);
      ctx[(size_t)b * ldc + c] = f2bf(a);
    }
  }
  return DIG_OK;
}

int dig_addattn_bwd(const void* xproj_, const void* sproj_, const float* w, const void* x_, const float* alpha, const void* dctx_, int ldd, float* dv,
                    void* dsproj_, float* dw_acc, int B, int N, int A, int X, hipStream_t) {
  if (!xproj_ || !sproj_ || !w || !x_ || !alpha || !dctx_ || !dv || !dsproj_ || !dw_acc || B <= 0 || N <= 0 || N > 512 || A <= 0 || A > 1024 || (A & 7) ||
      X <= 0 || X > 1024 || (X & 7))
    return DIG_ERR_ARG;
  if (!aligned16(xproj_) || !aligned16(x_)) return DIG_ERR_ALIGN;
  const bf16_t* xproj = (const bf16_t*)xproj_; const bf16_t* sproj = (const bf16_t*)sproj_; const bf16_t* x = (const bf16_t*)x_;
  const bf16_t* dctx = (const bf16_t*)dctx_;
  bf16_t* dsproj = (bf16_t*)dsproj_;
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    std::vector<float> g(N);
    float dot = 0.f;
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      for (int c = 0; c < X; ++c) acc += bf2f(dctx[(size_t)b * ldd + c]) * bf2f(x[((size_t)b * N + n) * X + c]);
      g[n] = acc;
      dot += alpha[(size_t)b * N + n] * acc;
    }
    for (int n = 0; n < N; ++n) { g[n] = alpha[(size_t)b * N + n] * (g[n] - dot); dv[(size_t)b * N + n] = g[n]; }
    for (int a = 0; a < A; ++a) {
      float ga = 0.f, wa = 0.f;
      const float sp = bf2f(sproj[(size_t)b * A + a]);
      for (int n = 0; n < N; ++n) {
        const float t = std::tanh(sp + bf2f(xproj[((size_t)b * N + n) * A + a]));
        ga += g[n] * (1.f - t * t);
        wa += g[n] * t;
      }
      dsproj[(size_t)b * A + a] = f2bf(ga * w[a]);
      dw_acc[(size_t)b * A + a] += wa;
    }
  }
  return DIG_OK;
}

int dig_addattn_bwd_tokens(const void* xproj_, const void* sproj_all_, const float* w, const float* dv_all, const float* alpha_all,
                           const void* dctx_all_, int ldd, void* dxproj_, void* dx_, int T, int B, int N, int A, int X, hipStream_t) {
  if (!xproj_ || !sproj_all_ || !w || !dv_all || !alpha_all || !dctx_all_ || !dxproj_ || !dx_ || T <= 0 || B <= 0 || N <= 0 || A <= 0 || (A & 1) || X <= 0 ||
      (X & 1))
    return DIG_ERR_ARG;
  if (((size_t)T * A + (size_t)T * 64) * 4 > 150 * 1024 || ((size_t)T * X + (size_t)T * 64) * 4 > 150 * 1024) return DIG_ERR_UNSUPPORTED;
  const bf16_t* xproj = (const bf16_t*)xproj_; const bf16_t* sproj_all = (const bf16_t*)sproj_all_; const bf16_t* dctx_all = (const bf16_t*)dctx_all_;
  bf16_t* dxproj = (bf16_t*)dxproj_; bf16_t* dx = (bf16_t*)dx_;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      for (int a = 0; a < A; ++a) {
        const float xv = bf2f(xproj[((size_t)b * N + n) * A + a]);
        float g = 0.f;
        for (int t = 0; t < T; ++t) {
          const float th = std::tanh(bf2f(sproj_all[((size_t)t * B + b) * A + a]) + xv);
          g += dv_all[((size_t)t * B + b) * N + n] * (1.f - th * th);
        }
        dxproj[((size_t)b * N + n) * A + a] = f2bf(g * w[a]);
      }
      for (int c = 0; c < X; ++c) {
        float g = 0.f;
        for (int t = 0; t < T; ++t) g += alpha_all[((size_t)t * B + b) * N + n] * bf2f(dctx_all[((size_t)t * B + b) * ldd + c]);
        dx[((size_t)b * N + n) * X + c] = f2bf(g);
      }
    }
  return DIG_OK;
}

int dig_gru_cell_fwd(const void* gi_, const void* gh_, const float* s_prev, float* s, void* s_bf16, float* gates, int B, int S, hipStream_t) {
  if (!gi_ || !gh_ || !s || !s_bf16 || !gates || B <= 0 || S <= 0) return DIG_ERR_ARG;
  const bf16_t* gi = (const bf16_t*)gi_; const bf16_t* gh = (const bf16_t*)gh_;
  bf16_t* sb = (bf16_t*)s_bf16;
  auto sig = [](float x) { return 1.f / (1.f + std::exp(-x)); };
  for (int i = 0; i < B * S; ++i) {
    const int b = i / S, k = i - b * S;
    const size_t g0 = (size_t)b * 3 * S + k;
    const float hn = bf2f(gh[g0 + 2 * S]);
    const float r = sig(bf2f(gi[g0]) + bf2f(gh[g0]));
    const float z = sig(bf2f(gi[g0 + S]) + bf2f(gh[g0 + S]));
    const float n = std::tanh(bf2f(gi[g0 + 2 * S]) + r * hn);
    const float sp = s_prev ? s_prev[i] : 0.f;
    const float v = (1.f - z) * n + z * sp;
    s[i] = v;
    sb[i] = f2bf(v);
    float* gt = gates + (size_t)b * 4 * S + k;
    gt[0] = r; gt[S] = z; gt[2 * S] = n; gt[3 * S] = hn;
  }
  return DIG_OK;
}

int dig_gru_cell_bwd(const float* ds_a, const float* ds_b, const float* ds_c, const float* ds_d, const float* gates, const float* s_prev, void* dgi_,
                     void* dgh_, float* ds_prev, int B, int S, hipStream_t) {
  if (!ds_a || !gates || !dgi_ || !dgh_ || !ds_prev || B <= 0 || S <= 0) return DIG_ERR_ARG;
  bf16_t* dgi = (bf16_t*)dgi_; bf16_t* dgh = (bf16_t*)dgh_;
  for (int i = 0; i < B * S; ++i) {
    const int b = i / S, k = i - b * S;
    float ds = ds_a[i];
    if (ds_b) ds += ds_b[i];
    if (ds_c) ds += ds_c[i];
    if (ds_d) ds += ds_d[i];
    const float* gt = gates + (size_t)b * 4 * S + k;
    const float r = gt[0], z = gt[S], n = gt[2 * S], hn = gt[3 * S];
    const float sp = s_prev ? s_prev[i] : 0.f;
    const float dn = ds * (1.f - z), dz = ds * (sp - n);
    const float dpre = dn * (1.f - n * n);
    const float dr = dpre * hn;
    const float gr = dr * r * (1.f - r), gz = dz * z * (1.f - z);
    const size_t g0 = (size_t)b * 3 * S + k;
    dgi[g0] = f2bf(gr); dgi[g0 + S] = f2bf(gz); dgi[g0 + 2 * S] = f2bf(dpre);
    dgh[g0] = f2bf(gr); dgh[g0 + S] = f2bf(gz); dgh[g0 + 2 * S] = f2bf(dpre * r);
    ds_prev[i] = ds * z;
  }
  return DIG_OK;
}

int dig_embed_rows(const long long* tokens, const float* table, void* out_, int ld, int rows, int cols, int vocab, hipStream_t) {
  if (!tokens || !table || !out_ || rows <= 0 || cols <= 0 || vocab <= 0 || ld < cols) return DIG_ERR_ARG;
  bf16_t* out = (bf16_t*)out_;
  for (int r = 0; r < rows; ++r) {
    const long long t = clamp_tok(tokens[r], vocab);
    for (int c = 0; c < cols; ++c) out[(size_t)r * ld + c] = f2bf(table[(size_t)t * cols + c]);
  }
  return DIG_OK;
}

}  // extern "C"
