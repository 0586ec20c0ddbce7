// ssw_gpu_cpp.h -- batched counterpart of the reference's C++ wrapper (mengyao/Complete-Striped-Smith-Waterman-Library
// src/ssw_cpp.h / ssw_cpp.cpp), header-only, over the batch C-ABI of libssw.so (include/ssw_gpu.h).
//
// The reference's StripedSmithWaterman::Aligner::Align (ssw_cpp.cpp:319-357) aligns ONE query per call: ssw_init + ssw_align +
// ConvertAlignment + CalculateNumberMismatch.  A caller that loops over reads keeps working unchanged on libssw.so (the drop-in
// path), but every call is a synchronous round trip to the GPU.  BatchAligner is the same operation for a whole vector of
// queries against the reference sequence(s) set once: ONE ssw_gpu_align_batch call, and the same Alignment records --
// sw_score, sw_score_next_best, ref/query begin/end, ref_end_next_best, mismatches, cigar / cigar_string with soft clips and
// '=' / 'X' runs -- as Aligner::Align returns for each query (tests/cpp/batch_check.cpp compares the two on the MI355X).
//
// Alignment and Filter are the reference's structs (ssw_cpp.h:15-63): when the reference header was included first they are
// used as they are; otherwise the same two structs are declared here, field for field, so that existing code compiles either way.
#ifndef SSW_GPU_CPP_H
#define SSW_GPU_CPP_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "ssw.h"
#include "ssw_gpu.h"

#ifndef COMPLETE_STRIPED_SMITH_WATERMAN_CPP_H_
namespace StripedSmithWaterman {
struct Alignment {
  uint16_t sw_score;
  uint16_t sw_score_next_best;
  int32_t ref_begin;
  int32_t ref_end;
  int32_t query_begin;
  int32_t query_end;
  int32_t ref_end_next_best;
  int32_t mismatches;
  std::string cigar_string;
  std::vector<uint32_t> cigar;
  Alignment() : sw_score(0), sw_score_next_best(0), ref_begin(0), ref_end(0), query_begin(0), query_end(0), ref_end_next_best(0), mismatches(0) {}
  void Clear() { *this = Alignment(); }
};
struct Filter {
  bool report_begin_position;
  bool report_cigar;
  uint16_t score_filter;
  uint16_t distance_filter;
  Filter() : report_begin_position(true), report_cigar(true), score_filter(0), distance_filter(32767) {}
  Filter(const bool& pos, const bool& cigar, const uint16_t& score, const uint16_t& dis)
      : report_begin_position(pos), report_cigar(cigar), score_filter(score), distance_filter(dis) {}
};
}  // namespace StripedSmithWaterman
#endif

namespace StripedSmithWaterman {

class BatchAligner {
 public:
  // default scoring of the reference's Aligner(): match 2, mismatch 2, gap open 3, gap extension 1, A/C/G/T/N
  BatchAligner(int device = 0) : ctx_(0), targets_(0), n_targets_(0), gap_open_(3), gap_extend_(1) {
    ctx_ = ssw_gpu_open(device);
    if (!ctx_) throw std::runtime_error(std::string("ssw_gpu_open: ") + ssw_gpu_last_error(0));
    SetDnaScores(2, 2);
  }
  BatchAligner(uint8_t match_score, uint8_t mismatch_penalty, uint8_t gap_opening_penalty, uint8_t gap_extending_penalty, int device = 0)
      : ctx_(0), targets_(0), n_targets_(0), gap_open_(gap_opening_penalty), gap_extend_(gap_extending_penalty) {
    ctx_ = ssw_gpu_open(device);
    if (!ctx_) throw std::runtime_error(std::string("ssw_gpu_open: ") + ssw_gpu_last_error(0));
    SetDnaScores(match_score, mismatch_penalty);
  }
  // user matrices, as Aligner(score_matrix, score_matrix_size, translation_matrix, translation_matrix_size) (ssw_cpp.cpp:216-231)
  BatchAligner(const int8_t* score_matrix, int score_matrix_size, const int8_t* translation_matrix, int translation_matrix_size, int device = 0)
      : ctx_(0), targets_(0), n_targets_(0), gap_open_(3), gap_extend_(1) {
    ctx_ = ssw_gpu_open(device);
    if (!ctx_) throw std::runtime_error(std::string("ssw_gpu_open: ") + ssw_gpu_last_error(0));
    matrix_size_ = score_matrix_size;
    matrix_.assign(score_matrix, score_matrix + score_matrix_size * score_matrix_size);
    table_.assign(128, 0);
    for (int i = 0; i < translation_matrix_size && i < 128; ++i) table_[i] = translation_matrix[i];
  }
  ~BatchAligner() {
    if (targets_) ssw_gpu_seqs_free(targets_);
    if (ctx_) ssw_gpu_close(ctx_);
  }
  void SetGapPenalty(uint8_t opening, uint8_t extending) { gap_open_ = opening; gap_extend_ = extending; }

  // the reference sequence(s), translated on the device and kept in HBM (Aligner::SetReferenceSequence keeps its translation on the host)
  int SetReferenceSequence(const char* seq, int length) {
    std::vector<std::string> one(1, std::string(seq, seq + length));
    return SetReferenceSequences(one);
  }
  int SetReferenceSequences(const std::vector<std::string>& refs) {
    if (targets_) { ssw_gpu_seqs_free(targets_); targets_ = 0; }
    std::string text; std::vector<int64_t> off(1, 0);
    for (size_t i = 0; i < refs.size(); ++i) { text += refs[i]; off.push_back((int64_t)text.size()); }
    targets_ = ssw_gpu_seqs_upload_ascii(ctx_, text.data(), off.data(), (int32_t)refs.size(), table_.data());
    if (!targets_) throw std::runtime_error(std::string("ssw_gpu_seqs_upload_ascii: ") + ssw_gpu_last_error(ctx_));
    ref_codes_.resize(text.size()); ref_off_ = off; n_targets_ = (int)refs.size();
    for (size_t i = 0; i < text.size(); ++i) ref_codes_[i] = table_[(unsigned char)text[i] & 127];
    return n_targets_;
  }

  // Every query against reference `target` (default: the first / only one): alignments[i] is what Aligner::Align(queries[i], filter,
  // &alignments[i], maskLen) gives; flags[i] (optional) its return value (s_align.flag).  maskLen < 15 is raised to 15 like there.
  void Align(const std::vector<std::string>& queries, const Filter& filter, std::vector<Alignment>* alignments, int32_t maskLen,
             std::vector<uint16_t>* flags = 0, int target = 0) const {
    if (!targets_ || target < 0 || target >= n_targets_) throw std::runtime_error("BatchAligner::Align: no such reference sequence");
    const int32_t nq = (int32_t)queries.size();
    alignments->assign(queries.size(), Alignment());
    if (flags) flags->assign(queries.size(), 0);
    if (nq == 0) return;
    std::string text; std::vector<int64_t> off(1, 0);
    for (int32_t i = 0; i < nq; ++i) { text += queries[i]; off.push_back((int64_t)text.size()); }
    ssw_gpu_seqs* Q = ssw_gpu_seqs_upload_ascii(ctx_, text.data(), off.data(), nq, table_.data());
    if (!Q) throw std::runtime_error(std::string("ssw_gpu_seqs_upload_ascii: ") + ssw_gpu_last_error(ctx_));
    ssw_gpu_params p; memset(&p, 0, sizeof p);
    p.mat = matrix_.data(); p.n = matrix_size_; p.gapO = gap_open_; p.gapE = gap_extend_;
    p.flag = (uint8_t)((filter.report_begin_position ? 0x08 : 0) | (filter.report_cigar ? 0x0f : 0));      // SetFlag, ssw_cpp.cpp:204-211
    p.filters = filter.score_filter; p.filterd = filter.distance_filter; p.maskLen = std::max(maskLen, 15); p.score_size = 2;
    std::vector<ssw_gpu_result> res((size_t)nq);
    uint32_t* pool = 0; int64_t words = 0;
    const int rc = ssw_gpu_align_batch(ctx_, Q, targets_, target, 1, &p, res.data(), &pool, &words);
    ssw_gpu_seqs_free(Q);
    if (rc != 0) { free(pool); throw std::runtime_error(std::string("ssw_gpu_align_batch: ") + (rc == SSW_GPU_BUSY ? ssw_gpu_strerror(rc) : ssw_gpu_last_error(ctx_))); }
    const int8_t* ref = ref_codes_.data() + ref_off_[(size_t)target];
    for (int32_t i = 0; i < nq; ++i) {
      const ssw_gpu_result& r = res[(size_t)i];
      Alignment& a = (*alignments)[(size_t)i];
      a.sw_score = r.score1; a.sw_score_next_best = r.score2; a.ref_begin = r.ref_begin1; a.ref_end = r.ref_end1;
      a.query_begin = r.read_begin1; a.query_end = r.read_end1; a.ref_end_next_best = r.ref_end2;
      if (flags) (*flags)[(size_t)i] = r.flag;
      const int qlen = (int)queries[(size_t)i].size();
      Expand(a, r.cigarLen > 0 ? pool + r.cigar_off : 0, r.cigarLen, ref, text.data() + off[(size_t)i], qlen);
    }
    free(pool);
  }

 private:
  // soft clips, '=' / 'X' runs and the mismatch count of one alignment: the outcome of the reference's ConvertAlignment +
  // CalculateNumberMismatch (ssw_cpp.cpp:52-89, 123-199).  Like there, the clips are written even when ssw_align returned no CIGAR
  // (CalculateNumberMismatch runs unconditionally: an alignment without a path ends up with just its soft clips).
  void Expand(Alignment& a, const uint32_t* cig, int32_t n_ops, const int8_t* ref, const char* query, int qlen) const {
    a.cigar.clear(); a.cigar_string.clear(); a.mismatches = 0;
    if (!cig) n_ops = 0;
    if (a.query_begin > 0) Push(a, (uint32_t)a.query_begin, 'S');
    const int8_t* rp = ref + (a.ref_begin > 0 ? a.ref_begin : 0);
    const char* qp = query + (a.query_begin > 0 ? a.query_begin : 0);
    uint32_t run = 0; char run_op = 0;
    for (int32_t k = 0; k < n_ops; ++k) {
      const char op = cigar_int_to_op(cig[k]);
      const uint32_t len = cigar_int_to_len(cig[k]);
      if (op == 'M') {
        for (uint32_t j = 0; j < len; ++j, ++rp, ++qp) {
          const char now = *rp != table_[(unsigned char)*qp & 127] ? 'X' : '=';
          if (now == 'X') ++a.mismatches;
          if (run > 0 && now != run_op) { Push(a, run, run_op); run = 0; }
          run_op = now; ++run;
        }
      } else if (op == 'I' || op == 'D') {
        if (run > 0) { Push(a, run, run_op); run = 0; }
        if (op == 'I') qp += len; else rp += len;
        a.mismatches += (int32_t)len;
        Push(a, len, op);
      }
    }
    if (run > 0) Push(a, run, run_op);
    const int end = qlen - a.query_end - 1;
    if (end > 0) Push(a, (uint32_t)end, 'S');
  }
  static void Push(Alignment& a, uint32_t len, char op) {
    a.cigar.push_back(to_cigar_int(len, op));
    a.cigar_string += std::to_string(len);
    a.cigar_string += op;
  }
  void SetDnaScores(uint8_t match, uint8_t mismatch) {      // BuildSwScoreMatrix + the A/C/G/T/N table of ssw_cpp.cpp:13-50 (N scores -mismatch)
    matrix_size_ = 5;
    matrix_.assign(25, (int8_t)-(int)mismatch);
    for (int i = 0; i < 4; ++i) matrix_[(size_t)(i * 5 + i)] = (int8_t)match;
    table_.assign(128, 4);
    table_['A'] = table_['a'] = 0; table_['C'] = table_['c'] = 1; table_['G'] = table_['g'] = 2; table_['T'] = table_['t'] = 3;
  }
  BatchAligner(const BatchAligner&);             // one device context per object
  BatchAligner& operator=(const BatchAligner&);

  ssw_gpu_ctx* ctx_;
  ssw_gpu_seqs* targets_;
  int n_targets_;
  uint8_t gap_open_, gap_extend_;
  int matrix_size_;
  std::vector<int8_t> matrix_, table_, ref_codes_;
  std::vector<int64_t> ref_off_;
};

}  // namespace StripedSmithWaterman
#endif  // SSW_GPU_CPP_H
