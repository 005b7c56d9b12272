// ggml-model.bin reader (reference loader biogpt.cpp:27-453,
// writer convert.py:28-97).  Parses the header, vocab, merges and the tensor directory;
// tensor payloads are read on demand so a 1.5 GB file never has to sit in host memory twice.
#pragma once

#include <map>
#include <string>
#include <vector>

#include "host_common.h"

namespace bg {

struct TensorEntry {
    std::string name;
    int32_t type = T_INVALID;  // ggml_type id
    int32_t n_dims = 0;
    int64_t ne0 = 1;           // innermost dimension (row length)
    int64_t ne1 = 1;           // number of rows
    uint64_t file_offset = 0;  // payload offset in the file
    size_t nbytes = 0;         // payload bytes (file layout)
};

struct ModelFile {
    std::string path;
    biogpt_hip_hparams hp{};
    std::vector<std::string> vocab;   // id -> token bytes (biogpt.cpp:72-113)
    std::vector<std::string> merges;  // rank -> "left right" (biogpt.cpp:116-156)
    std::vector<TensorEntry> tensors; // file order
    std::map<std::string, int> by_name;

    // Parse `fname`; returns false (error set) on any of the reference's load failures.
    bool open(const std::string &fname);
    const TensorEntry *find(const std::string &name) const;
    // Read one tensor's payload (file layout) into dst (nbytes).
    bool read_payload(const TensorEntry &t, void *dst) const;
};

// The tensor names the loader binds (biogpt.cpp:258-317) with their expected shapes.
struct ExpectedTensor {
    std::string name;
    int64_t ne0, ne1;  // ne1 < 0: taken from the file (embed_positions, F5)
    bool matrix;       // true: carries the file's weight type; false: F32 vector
};
std::vector<ExpectedTensor> expected_tensors(const biogpt_hip_hparams &hp);

}  // namespace bg
