// grim_tok.h -- what the device GL tokenizer (grim_tokdev.h) and its host side (grim_engine.hip, grim_stream.cpp,
// grim_host.cpp) share: the per-line record the host hands over, the device form of the allele dictionary, and the hash
// both sides put names through.  Plain data, no HIP headers.
//
// What moves to the device: for the by far most common kind of input line -- every locus typed with ONE allele per side,
// no '/' list (BASELINE configs 2 and 3, the bulk of a registry file) -- clean_up_gl / gl2haps (impute.py:105-118,
// 246-272), i.e. the work of tokenise_gl_fast (grim_host.cpp).  The host keeps what is cheap and serial: cutting the text
// into lines and fields (impute.py:2022-2036), the race pair -> prior matrix index, and the one-byte test "the GL field has
// no '/'" that makes a line a candidate.  A candidate the device cannot take (an allele the graph does not know, 'g' / 'L'
// characters for clean_up_gl to remove, loci out of order, a missing locus ...) is flagged and goes through the host's
// general tokenizer afterwards (grim_stream.cpp: fix-up pass); the result is the same either way.
#pragma once
#include <stdint.h>

#include "../../include/grim_hip.h"

struct LineRec {       // one input line as the host's line splitter leaves it
  uint32_t gl_off;     // the GL field: offset into the chunk's text ...
  uint16_t gl_len;     // ... and length; 0 = not a line for the device tokenizer
  uint16_t prior_idx;  // index of the line's (race1, race2) prior matrix
};                     // 8 bytes

#define GRIM_TOKNAME 24  // allele names up to this many bytes are in the device dictionary (longer ones: host tokenizer)
#define GRIM_TOK_MAXGL 144  // longest GL field the device tokenizer looks at (a 160-byte window from the 16-byte boundary before it)

struct DictEnt {       // one slot of a locus's open-addressing table: the name inline, so a probe is two 16-byte loads
  uint32_t w[6];       // name bytes packed BIG-endian, zero padded: comparing the words in order compares the strings
  uint32_t meta;       // bit 31: used; bits 16..23: length; bits 0..15: allele id
  uint32_t hash;
};                     // 32 bytes

struct DevDict {
  const DictEnt *tab[GRIM_MAXL];
  uint32_t mask[GRIM_MAXL];
  uint64_t locus[GRIM_MAXL];    // locus name (the bytes before '*') packed big-endian, zero padded; at most 8 bytes
  uint32_t locus_len[GRIM_MAXL];
  uint32_t n_loci;
};

// hash of a packed name (both sides build and probe the tables with it)
#if defined(__HIPCC__) || defined(__CUDACC__)
#define GRIM_HD __host__ __device__
#else
#define GRIM_HD
#endif
GRIM_HD static inline uint32_t tokname_hash(const uint32_t (&w)[6], uint32_t len) {
  uint32_t h = len * 0x9E3779B1u;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    h ^= w[i];
    h *= 0x85EBCA6Bu;
    h ^= h >> 15;
  }
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
