/* salmetric_oracle.c -- TEST INFRASTRUCTURE ONLY: plain-C restatement of the reference's saliency metrics
 * (CSNet_training/SalMetric/src/sal_metric.cpp) on raw 8-bit arrays.  The reference binary itself needs OpenCV
 * (imread / Mat), which this image does not have, so it cannot be built here ("unbuildable"); the loops
 * below follow its arithmetic statement by statement, in float like the reference:
 *   compute_mae                 sal_metric.cpp:87-97    mae += abs(sal - gt) / 255. per pixel, then / (h * w)
 *   compute_precision_and_recall sal_metric.cpp:99-120  per threshold th in 0..255: a = sal > th, b = gt > 128,
 *                                                        pre = (ab + 1e-4) / (a_sum + 1e-4), rec likewise with b_sum
 * Only tests/ may load this library (oracle/Makefile builds it); the product path is csn_sal_hist + metric.py. */
#include <math.h>
#include <stdint.h>

#define THRESHOLDS 256
static const float EPS = 1e-4f;

float salm_mae(const uint8_t* sal, const uint8_t* gt, int height, int width) {
  float mae = 0;
  for (int h = 0; h < height; ++h)
    for (int w = 0; w < width; ++w) {
      const float s = (float)sal[h * width + w], g = (float)gt[h * width + w];
      mae += fabsf(s - g) / 255.;   /* double division, rounded back to float by the += */
    }
  return mae / (height * width);
}

/* adds this image's precision / recall per threshold to the two arrays (like the reference's per-thread sums) */
void salm_precision_recall(const uint8_t* sal, const uint8_t* gt, int height, int width, float* precision, float* recall) {
  for (int th = 0; th < THRESHOLDS; ++th) {
    float a_sum = 0, b_sum = 0, ab = 0;
    for (int i = 0; i < height * width; ++i) {
      const unsigned a = ((float)sal[i] > th) ? 1 : 0;
      const unsigned b = ((float)gt[i] > THRESHOLDS / 2) ? 1 : 0;
      ab += (a & b);
      a_sum += a;
      b_sum += b;
    }
    precision[th] += (ab + EPS) / (a_sum + EPS);
    recall[th] += (ab + EPS) / (b_sum + EPS);
  }
}
