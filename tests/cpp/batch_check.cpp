// tests/cpp/batch_check.cpp -- TEST INFRASTRUCTURE.  The reference's own C++ wrapper (src/ssw_cpp.cpp, compiled from where it
// lies under /root/reference by oracle/Makefile `batchcpp`) calling libssw.so pair by pair, against BatchAligner
// (include/ssw_gpu_cpp.h) on the same seeded reads in ONE batch call: every Alignment field, cigar vector and string, return flag.
// Usage: batch_cpp_check [reads] [read_len] [ref_len]   -> prints "ok <n>" or the first differences; exit code 0 / 1.
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "ssw_cpp.h"        // the reference's header (-I/root/reference/src at build time)
#include "ssw_gpu_cpp.h"

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }

int main(int argc, char** argv)
{
	const int nreads = argc > 1 ? atoi(argv[1]) : 300, rlen = argc > 2 ? atoi(argv[2]) : 120, reflen = argc > 3 ? atoi(argv[3]) : 5000;
	const char acgt[] = "ACGTN";
	std::string ref((size_t)reflen, 'A');
	for (int i = 0; i < reflen; ++i) ref[(size_t)i] = acgt[rnd() % 400 == 0 ? 4 : rnd() % 4];
	std::vector<std::string> reads;
	for (int r = 0; r < nreads; ++r) {
		std::string s;
		const int L = 20 + (int)(rnd() % (unsigned)rlen);
		if (rnd() % 10 == 0) for (int i = 0; i < L; ++i) s += acgt[rnd() % 4];      // unrelated read
		else {
			int p = (int)(rnd() % (unsigned)(reflen - L - 20));
			while ((int)s.size() < L) {
				const unsigned e = rnd() % 100;
				if (e < 3) s += acgt[rnd() % 4], ++p;          // substitution
				else if (e < 4) s += acgt[rnd() % 4];          // insertion
				else if (e < 5) ++p;                           // deletion
				else s += ref[(size_t)p++];
			}
		}
		reads.push_back(s);
	}
	int bad = 0, total = 0;
	for (int mode = 0; mode < 3; ++mode) {
		StripedSmithWaterman::Filter filter;
		if (mode == 1) { filter.report_cigar = false; }                                     // begin positions only
		if (mode == 2) { filter.report_cigar = false; filter.report_begin_position = false; }   // scores only
		if (mode == 0) { filter.score_filter = 60; filter.distance_filter = 110; }          // some alignments filtered out of the CIGAR stage
		const int32_t maskLen = mode == 1 ? 7 : 40;       // (below 15: raised to 15 by both)
		StripedSmithWaterman::Aligner one(2, 2, 3, 1);
		StripedSmithWaterman::BatchAligner many(2, 2, 3, 1);
		many.SetReferenceSequence(ref.data(), (int)ref.size());
		std::vector<StripedSmithWaterman::Alignment> got;
		std::vector<uint16_t> gflags;
		many.Align(reads, filter, &got, maskLen, &gflags);
		for (int r = 0; r < nreads; ++r, ++total) {
			StripedSmithWaterman::Alignment e;
			const uint16_t ef = one.Align(reads[(size_t)r].c_str(), reads[(size_t)r].size(), ref.c_str(), ref.size(), filter, e, maskLen);
			const StripedSmithWaterman::Alignment& g = got[(size_t)r];
			const bool same = e.sw_score == g.sw_score && e.sw_score_next_best == g.sw_score_next_best && e.ref_begin == g.ref_begin && e.ref_end == g.ref_end &&
			                  e.query_begin == g.query_begin && e.query_end == g.query_end && e.ref_end_next_best == g.ref_end_next_best &&
			                  e.mismatches == g.mismatches && e.cigar_string == g.cigar_string && e.cigar == g.cigar && ef == gflags[(size_t)r];
			if (!same && ++bad <= 5)
				printf("mode %d read %d: expected %u %u %d %d %d %d %d mm %d '%s' flag %u, got %u %u %d %d %d %d %d mm %d '%s' flag %u\n", mode, r,
				       e.sw_score, e.sw_score_next_best, e.ref_begin, e.ref_end, e.query_begin, e.query_end, e.ref_end_next_best, e.mismatches, e.cigar_string.c_str(), ef,
				       g.sw_score, g.sw_score_next_best, g.ref_begin, g.ref_end, g.query_begin, g.query_end, g.ref_end_next_best, g.mismatches, g.cigar_string.c_str(), gflags[(size_t)r]);
		}
	}
	if (bad) { printf("FAILED: %d of %d alignments differ\n", bad, total); return 1; }
	printf("ok %d\n", total);
	return 0;
}
