/* context_sampler.c -- TEST INFRASTRUCTURE (oracle), not product code.
 *
 * CPU restatement of the device context sampler / verifier (include/carl_amd.h:
 * carl_sample_contexts, carl_verify_contexts), i.e. of what replaces
 * ContextSampler.sample_contexts (carl/context/sampler.py:45-61) + the default fill
 * (carl/envs/carl_env.py:135-137) and ContextSpace.verify_context
 * (carl/context/context_space.py:54-59) for dense context sets.
 *
 * The device sampler is a NEW random stream (Philox keyed by seed / context id / feature), not
 * the reference's NumPy MT19937 one: the reference's recorded sampler outputs pin the HOST
 * sampler (carl_amd/context/sampler.py, tests/golden/notebook_sampler_outputs.json); what can be
 * pinned here is the distribution family per feature type (tests/test_device_sampler.py) and the
 * device kernel against this file.  Uniform / integer / categorical / constant draws are the
 * same float32 expressions as the kernel (bit-exact); the normal draw is evaluated in double
 * and rounded once (the kernel's float32 log/sqrt/sincos differ in the last bits).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../include/carl_amd.h"

void oracle_lane_words(uint64_t seed, uint64_t glane, uint32_t episode, uint32_t sub, uint32_t out[4]);
float oracle_u01(uint32_t w);

#define SUB_SAMPLER 0x40000000u
#define NORMAL_TRIES 32

static float sample_feature(const carl_feature_spec_t* sp, uint64_t seed, uint64_t gctx, uint32_t f) {
  if (sp->kind == CARL_FEAT_CONSTANT) return sp->value;
  uint32_t w[4];
  oracle_lane_words(seed, gctx, f, SUB_SAMPLER, w);
  const float u = oracle_u01(w[0]);
  if (sp->kind == CARL_FEAT_UNIFORM_FLOAT) {
    if (sp->log_scale) {
      const double lo = log((double)sp->lower), hi = log((double)sp->upper);
      return (float)exp(lo + (hi - lo) * (double)u);
    }
    return fmaf(sp->upper - sp->lower, u, sp->lower);
  }
  if (sp->kind == CARL_FEAT_UNIFORM_INT) {
    const float span = sp->upper - sp->lower + 1.0f;
    float k = floorf(u * span);
    if (k > span - 1.0f) k = span - 1.0f;
    return sp->lower + k;
  }
  if (sp->kind == CARL_FEAT_CATEGORICAL) {
    int k = (int)(u * (float)sp->n_choices);
    if (k > sp->n_choices - 1) k = sp->n_choices - 1;
    return sp->choices[k];
  }
  float v = 0.0f;
  for (int attempt = 0; attempt < NORMAL_TRIES; ++attempt) {
    if (attempt > 0) oracle_lane_words(seed, gctx, f, SUB_SAMPLER | (uint32_t)attempt, w);
    const double u1 = (double)oracle_u01(w[0]), u2 = (double)oracle_u01(w[1]);
    const double z = sqrt(-2.0 * log(1.0 - u1)) * cos(6.28318530717958647692 * u2);
    v = (float)((double)sp->mu + (double)sp->sigma * z);
    if (v >= sp->lower && v <= sp->upper) return v;
  }
  if (v < sp->lower) v = sp->lower;
  if (v > sp->upper) v = sp->upper;
  return v;
}

void oracle_sample_contexts(const carl_feature_spec_t* specs, int n_features, int n_contexts, int ctx_stride,
                            int64_t context_offset, uint64_t seed, float* ctx_table) {
  for (int f = 0; f < n_features; ++f)
    for (int c = 0; c < n_contexts; ++c)
      ctx_table[(size_t)f * ctx_stride + c] = sample_feature(&specs[f], seed, (uint64_t)(context_offset + c), (uint32_t)f);
}

int oracle_verify_contexts(const carl_feature_spec_t* specs, int n_features, int n_contexts, int ctx_stride,
                           const float* ctx_table) {
  int bad = 0;
  for (int f = 0; f < n_features; ++f)
    for (int c = 0; c < n_contexts; ++c) {
      const float v = ctx_table[(size_t)f * ctx_stride + c];
      int ok;
      if (specs[f].kind == CARL_FEAT_CATEGORICAL) {
        ok = 0;
        for (int k = 0; k < specs[f].n_choices; ++k) ok |= (v == specs[f].choices[k]);
      } else {
        ok = (v >= specs[f].lower) && (v <= specs[f].upper);
      }
      bad += !ok;
    }
  return bad;
}
