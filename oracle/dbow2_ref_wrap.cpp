// oracle/_ref/libdbow2_vec.so -- TEST INFRASTRUCTURE ONLY.
// A C entry point over the REFERENCE'S OWN DBoW2::BowVector / DBoW2::FeatureVector.  This file holds no reference
// code: oracle/Makefile compiles it together with /root/reference/Thirdparty/DBoW2/DBoW2/BowVector.cpp and
// FeatureVector.cpp (plain g++ on those two files where they lie; both include only the standard library), output
// into oracle/_ref/ (git-ignored).  The oracle's orc_bow_transform accumulates through it after orc_use_dbow2_ref(),
// which pins the double arithmetic of Frame::ComputeBoW's BowVector (addWeight, addIfNotExist, normalize) and the
// FeatureVector's grouping (addFeature) to reference code.  The calls mirror
// TemplatedVocabulary<>::transform(features, v, fv, levelsup), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1204
// (that header itself needs OpenCV and cannot be compiled here; the descent it performs is restated in the oracle).
#include <stdint.h>
#include "BowVector.h"
#include "FeatureVector.h"

extern "C" int dbow2ref_accumulate(const uint32_t* word, const double* weight, const uint32_t* node, int n,
                                   int weighting, int must, int norm_l2, uint32_t* bow_ids, double* bow_vals,
                                   int* n_words, uint32_t* fv_nodes, uint32_t* fv_off, uint32_t* fv_feat, int* n_fv) {
  DBoW2::BowVector v;
  DBoW2::FeatureVector fv;
  const bool tf = weighting == DBoW2::TF || weighting == DBoW2::TF_IDF;
  for (int i = 0; i < n; i++) {
    if (!(weight[i] > 0)) continue;                      // stopped word (:1167, :1194)
    if (tf) v.addWeight(word[i], weight[i]);             // :1169
    else v.addIfNotExist(word[i], weight[i]);            // :1196
    fv.addFeature(node[i], (unsigned)i);                 // :1170, :1197
  }
  if (tf && !v.empty() && !must) {                       // :1174-1180
    const double nd = v.size();
    for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
  }
  if (must) v.normalize(norm_l2 ? DBoW2::L2 : DBoW2::L1);   // :1203
  int nw = 0;
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it) {
    bow_ids[nw] = it->first;
    bow_vals[nw] = it->second;
    nw++;
  }
  *n_words = nw;
  int nn = 0, pos = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
    fv_nodes[nn] = it->first;
    fv_off[nn] = pos;
    for (size_t j = 0; j < it->second.size(); j++) fv_feat[pos++] = it->second[j];
    nn++;
  }
  fv_off[nn] = pos;
  *n_fv = nn;
  return 0;
}
