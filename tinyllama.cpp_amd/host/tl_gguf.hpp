// tl_gguf.hpp — GGUF v3 reader of the C++ host layer: maps the file and exposes metadata + raw tensor bytes, so that
// quantized super-blocks go from the page cache to the device without a typed host copy.
//
// Replaces gguf_parser.cpp:123-806 (load_gguf_meta: typed std::vector copies of every tensor) for the new backend.
// Format facts as the reference parses them: little-endian; header magic "GGUF", u32 version,
// u64 tensor count, u64 kv count; strings are u64 length + bytes; tensor info = name, u32 n_dims (<= 4), u64 dims,
// u32 ggml type, u64 offset relative to the data section, which starts at the next multiple of general.alignment
// (default 32). dims[0] = columns (in_features), dims[1] = rows.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace tlhost {

enum GGMLType : uint32_t { GGML_F32 = 0, GGML_F16 = 1, GGML_Q8_0 = 8, GGML_Q4_K = 12, GGML_Q6_K = 14, GGML_Q8_K = 15, GGML_BF16 = 30 };

// bytes / elements of one block of a ggml type this backend can hold (quantization.h:57-124); false if unknown
bool ggml_block(uint32_t type, size_t* block_bytes, size_t* block_elems);

struct GGUFValue {
  enum Kind { NONE, UINT, INT, FLOAT, BOOL, STRING, ARRAY } kind = NONE;
  uint64_t u = 0;
  int64_t i = 0;
  double f = 0.0;
  std::string s;
  uint32_t array_type = 0;   // ARRAY: element type and count (elements are skipped, except string arrays)
  uint64_t array_count = 0;
  std::vector<std::string> strings;  // ARRAY of strings (tokenizer.ggml.tokens / merges)
  std::vector<float> floats;         // tokenizer.ggml.scores (F32 array), gguf_parser.cpp:255-266
  std::vector<int32_t> ints;         // tokenizer.ggml.token_type (U32 or I32 array), gguf_parser.cpp:268-292
  bool as_integer(int64_t* out) const;  // any integral kind (model_config.cpp:89-147 accepts every integral type)
  bool as_double(double* out) const;
};

struct GGUFTensor {
  std::string name;
  uint32_t type = 0;
  std::vector<uint64_t> dims;
  uint64_t offset = 0;      // relative to the data section
  size_t nbytes = 0;
  const uint8_t* data = nullptr;  // into the mapping
  uint64_t cols() const { return dims.empty() ? 0 : dims[0]; }
  uint64_t rows() const { return dims.size() < 2 ? 1 : dims[1]; }
};

class GGUFFile {
 public:
  GGUFFile() = default;
  ~GGUFFile();
  GGUFFile(const GGUFFile&) = delete;
  GGUFFile& operator=(const GGUFFile&) = delete;
  // throws std::runtime_error with the reason (bad magic, truncated file, unknown type, size not a block multiple ...)
  void open(const std::string& path);
  void close();
  uint32_t version() const { return version_; }
  uint64_t alignment() const { return alignment_; }
  const std::map<std::string, GGUFValue>& metadata() const { return meta_; }
  const std::vector<GGUFTensor>& tensors() const { return tensors_; }
  const GGUFTensor* find(const std::string& name) const;
  const GGUFValue* meta(const std::string& key) const;
  size_t file_bytes() const { return size_; }

 private:
  int fd_ = -1;
  const uint8_t* map_ = nullptr;
  size_t size_ = 0;
  uint32_t version_ = 0;
  uint64_t alignment_ = 32;
  std::map<std::string, GGUFValue> meta_;
  std::vector<GGUFTensor> tensors_;
};

}  // namespace tlhost
