// grim_host_internal.h -- shared by the host-side translation units of the library (no GPU code)
#pragma once
#include <stdint.h>

#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/grim_hip.h"

using sv = std::string_view;

// allele dictionary: per locus slot, allele string <-> dense id
struct grim_dict {
  uint32_t n_loci;
  std::vector<std::string> locus_name;
  std::unordered_map<std::string, uint32_t> locus_slot;
  std::vector<std::unordered_map<std::string, uint32_t>> ids;
  std::vector<std::vector<std::string>> names;
};

int32_t dict_intern(grim_dict *d, uint32_t slot, sv a);  // -1 when the locus has run out of ids
void py_float(double x, std::string &out);               // CPython str(float)
