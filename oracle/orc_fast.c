/*
 * orc_fast.c -- CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED, see
 * vo_oracle.h.
 *
 * cv::FAST(image, keypoints, 20, true) as called by featureDetectionFast
 * (src/feature.cpp:39-47): FastFeatureDetector::TYPE_9_16, restated from OpenCV 4.5.x
 * features2d/src/fast.cpp (FAST_t<16>) and fast_score.cpp (cornerScore<16>) -- SURVEY App. A7.
 * Output order is row-major scan order (bucketing is order dependent, quirk B2').
 */
#include "orc_internal.h"

#include <string.h>

static const int OFFS16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1},
                                  {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                  {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

/* returns 0 for non-corners, else (uchar)cornerScore<16> */
static int fast_score_at(const uint8_t *p, const int *pixel, int threshold)
{
    int v = p[0], d[25];
    for (int k = 0; k < 25; k++)
        d[k] = v - p[pixel[k]];
    /* is it a corner at `threshold`?  >= 9 contiguous strictly darker or strictly brighter */
    int is_corner = 0;
    int count = 0;
    for (int k = 0; k < 25; k++) {
        if (d[k] > threshold) {
            if (++count > 8) {
                is_corner = 1;
                break;
            }
        } else
            count = 0;
    }
    if (!is_corner) {
        count = 0;
        for (int k = 0; k < 25; k++) {
            if (-d[k] > threshold) {
                if (++count > 8) {
                    is_corner = 1;
                    break;
                }
            } else
                count = 0;
        }
    }
    if (!is_corner)
        return -1;
    /* cornerScore<16> */
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0)
            continue;
        for (int q = 4; q <= 8; q++)
            a = a < d[k + q] ? a : d[k + q];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int q = 3; q <= 5; q++)
            b = b > d[k + q] ? b : d[k + q];
        if (b >= b0)
            continue;
        for (int q = 6; q <= 8; q++)
            b = b > d[k + q] ? b : d[k + q];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return (uint8_t)(-b0 - 1);
}

int orc_fast_detect(const uint8_t *img, int w, int h, int threshold, int nonmax, float *pts, int cap)
{
    int pixel[25];
    for (int k = 0; k < 16; k++)
        pixel[k] = OFFS16[k][0] + OFFS16[k][1] * w;
    for (int k = 16; k < 25; k++)
        pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;

    uint8_t *score = (uint8_t *)calloc((size_t)w * h, 1);
    uint8_t *corner = (uint8_t *)calloc((size_t)w * h, 1);
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            int s = fast_score_at(img + (size_t)i * w + j, pixel, threshold);
            if (s >= 0) {
                corner[(size_t)i * w + j] = 1;
                score[(size_t)i * w + j] = (uint8_t)s;
            }
        }
    int n = 0;
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            if (!corner[(size_t)i * w + j])
                continue;
            int sc = score[(size_t)i * w + j];
            const uint8_t *prev = score + (size_t)i * w, *pprev = prev - w, *curr = prev + w;
            if (!nonmax || (sc > prev[j + 1] && sc > prev[j - 1] && sc > pprev[j - 1] &&
                            sc > pprev[j] && sc > pprev[j + 1] && sc > curr[j - 1] &&
                            sc > curr[j] && sc > curr[j + 1])) {
                if (n < cap) {
                    pts[2 * n] = (float)j;
                    pts[2 * n + 1] = (float)i;
                }
                n++;
            }
        }
    free(score);
    free(corner);
    return n;
}
