/*
 * orbv.h -- C ABI of the bag-of-words assignment that precedes SearchByBow / SearchForTriangulation (liborbx.so).
 *
 * Replaces Frame::computeBow / KeyFrame::computeBow (modules/BasicObject/Frame.cpp:168-178), i.e.
 * DBoW2::TemplatedVocabulary<FORB>::transform(features, BowVector&, FeatureVector&, levelsup = 4)
 * (thirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259): every descriptor walks the k-ary vocabulary tree by
 * minimum Hamming distance (first child wins ties, :1241), the words' weights are accumulated into the BowVector
 * (BowVector.cpp:32-45) and normalised (BowVector.cpp:62-90), and the feature indices are grouped by their ancestor
 * node `levelsup` levels above the leaves (FeatureVector.cpp:31-45).  SURVEY.md 8f rank 3.
 *
 * The tree lives in HBM for the lifetime of the handle (ORBvoc: k = 10, L = 6, ~1.1 M nodes x 32 B = 35 MB); the
 * outputs are the two std::maps flattened in key order -- the FeatureVector in exactly the CSR form orbm_fv takes.
 *
 * Returns 0 or a negative ORBX_E_* code (orbx.h); text in orbx_last_error().
 * Threads (orbx.h, "Streams and threads"): the reference's vocabulary is ONE object that Tracking (Frame.cpp:168-178) and
 * LocalMapping (LocalMapping.cpp:90) call at the same time.  The tree is read-only; orbv_transform (host pointers) leases a
 * stream, scratch and staging of its own per call, so it is RE-ENTRANT on one shared handle (tests/cpp/two_threads.cpp).  The
 * *_device entry points use the handle's one per-feature scratch: keep the device transform calls of one handle on one stream
 * (they are then ordered); orbv_transform_features_device uses no scratch and may run anywhere.
 */
#ifndef ORBV_H
#define ORBV_H

#include <stddef.h>
#include <stdint.h>

#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* DBoW2::WeightingType / ScoringType (BowVector.h:39-56) */
enum { ORBV_TF_IDF = 0, ORBV_TF = 1, ORBV_IDF = 2, ORBV_BINARY = 3 };
enum { ORBV_L1_NORM = 0, ORBV_L2_NORM = 1, ORBV_CHI_SQUARE = 2, ORBV_KL = 3, ORBV_BHATTACHARYYA = 4, ORBV_DOT_PRODUCT = 5 };

#define ORBV_MAX_FEATURES 8192   /* per frame, limit of the grouping kernel's LDS sort */
#define ORBV_NO_NODE 0xFFFFFFFFu /* node id of a feature whose descent ended above level L - levelsup: the reference
                                    leaves `nid` uninitialised there (TemplatedVocabulary.h:1150,1228-1250) */

typedef struct orbv_ctx orbv_t;

/* Vocabulary from plain arrays, nodes in the order loadFromTextFile creates them (TemplatedVocabulary.h:1376-1417):
 * node 0 is the root, parent[i] < i, children keep ascending-id order, word ids count the is_leaf flags in order. */
int orbv_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                const uint8_t *desc /* n_nodes x 32 */, const double *weight, int device, orbv_t **out);
/* TemplatedVocabulary::loadFromTextFile (:1338-1420), the ORBvoc.txt format: "k L scoring weighting" then one line
 * "parent is_leaf d0 .. d31 weight" per node.  The empty last line, which the reference's eof loop turns into a
 * phantom child of the root with an uninitialised descriptor, is NOT turned into a node. */
int orbv_load_text(const char *path, int device, orbv_t **out);
void orbv_destroy(orbv_t *h);
int orbv_info(const orbv_t *h, int *k, int *L, int *scoring, int *weighting, int *n_nodes, int *n_words);
/* copies the parsed tree back (for checking a loader): any pointer may be NULL */
int orbv_nodes(const orbv_t *h, int32_t *parent, uint8_t *is_leaf, uint8_t *desc, double *weight);

/* transform(feature, word_id, weight, &nid, levelsup) (:1218-1259) for n descriptors.  Device pointers, enqueued on `stream`
 * (hipStream_t; NULL: orbx.h, "Streams"). */
int orbv_transform_features_device(orbv_t *h, const uint8_t *d_desc, int n, int levelsup, uint32_t *d_word,
                                   uint32_t *d_node, double *d_weight, void *stream);

/* transform(features, BowVector, FeatureVector, levelsup) (:1127-1201) for a batch of frames.  Device pointers:
 *   d_desc [n_frames][cap][32], d_n [n_frames] (each <= cap; cap > ORBV_MAX_FEATURES is refused with
 *   ORBX_E_UNSUPPORTED, a count above cap is clamped to cap as orbx_extract_batch_device clamps its records)
 *   d_bow_ids / d_bow_vals [n_frames][cap], d_n_words [n_frames]          BowVector, ascending word id
 *   d_fv_nodes [n_frames][cap], d_fv_off [n_frames][cap + 1], d_fv_idx [n_frames][cap], d_n_fv [n_frames]
 *                                                                        FeatureVector as CSR, ascending node id,
 *                                                                        feature indices ascending inside a node
 * Enqueued on `stream` (NULL: orbx.h, "Streams"); no host synchronisation. */
int orbv_transform_device(orbv_t *h, int n_frames, const uint8_t *d_desc, const int32_t *d_n, int cap, int levelsup,
                          uint32_t *d_bow_ids, double *d_bow_vals, int32_t *d_n_words, uint32_t *d_fv_nodes,
                          int32_t *d_fv_off, uint32_t *d_fv_idx, int32_t *d_n_fv, void *stream);
/* one frame, host pointers; arrays sized n (fv_off n + 1) */
int orbv_transform(orbv_t *h, const uint8_t *desc, int n, int levelsup, uint32_t *bow_ids, double *bow_vals,
                   int32_t *n_words, uint32_t *fv_nodes, int32_t *fv_off, uint32_t *fv_idx, int32_t *n_fv);

#ifdef __cplusplus
}
#endif
#endif
