// tl_gguf.cpp — see tl_gguf.hpp.
#include "tl_gguf.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <stdexcept>

namespace tlhost {

bool ggml_block(uint32_t type, size_t* bb, size_t* be) {
  switch (type) {
    case GGML_F32: *bb = 4; *be = 1; return true;
    case GGML_F16: *bb = 2; *be = 1; return true;
    case GGML_BF16: *bb = 2; *be = 1; return true;
    case GGML_Q8_0: *bb = 34; *be = 32; return true;
    case GGML_Q4_K: *bb = 144; *be = 256; return true;
    case GGML_Q6_K: *bb = 210; *be = 256; return true;
    case GGML_Q8_K: *bb = 290; *be = 256; return true;  // (the reference's parser rejects Q8_K files: block size 0)
    default: return false;
  }
}

bool GGUFValue::as_integer(int64_t* out) const {
  if (kind == UINT) *out = (int64_t)u;
  else if (kind == INT) *out = i;
  else if (kind == BOOL) *out = u ? 1 : 0;
  else return false;
  return true;
}
bool GGUFValue::as_double(double* out) const {
  if (kind == FLOAT) *out = f;
  else if (kind == UINT) *out = (double)u;
  else if (kind == INT) *out = (double)i;
  else return false;
  return true;
}

namespace {

// bounds-checked little-endian cursor over the mapping
struct Cursor {
  const uint8_t* p;
  size_t size, pos = 0;
  void need(size_t n) const {
    if (n > size || pos > size - n) throw std::runtime_error("GGUF: file truncated at byte " + std::to_string(pos));
  }
  template <typename T>
  T get() {
    need(sizeof(T));
    T v;
    memcpy(&v, p + pos, sizeof(T));
    pos += sizeof(T);
    return v;
  }
  std::string str() {
    const uint64_t n = get<uint64_t>();
    if (n > (1u << 30)) throw std::runtime_error("GGUF: unreasonable string length");
    need((size_t)n);
    std::string s(reinterpret_cast<const char*>(p + pos), (size_t)n);
    pos += (size_t)n;
    return s;
  }
  void skip(size_t n) {
    need(n);
    pos += n;
  }
};

// GGUF metadata value types (ggml_types.h:51-65)
enum : uint32_t { VT_U8 = 0, VT_I8, VT_U16, VT_I16, VT_U32, VT_I32, VT_F32, VT_BOOL, VT_STRING, VT_ARRAY, VT_U64, VT_I64, VT_F64 };

size_t scalar_size(uint32_t t) {
  switch (t) {
    case VT_U8: case VT_I8: case VT_BOOL: return 1;
    case VT_U16: case VT_I16: return 2;
    case VT_U32: case VT_I32: case VT_F32: return 4;
    case VT_U64: case VT_I64: case VT_F64: return 8;
    default: return 0;
  }
}

GGUFValue read_scalar(Cursor& c, uint32_t t) {
  GGUFValue v;
  switch (t) {
    case VT_U8: v.kind = GGUFValue::UINT; v.u = c.get<uint8_t>(); break;
    case VT_I8: v.kind = GGUFValue::INT; v.i = c.get<int8_t>(); break;
    case VT_U16: v.kind = GGUFValue::UINT; v.u = c.get<uint16_t>(); break;
    case VT_I16: v.kind = GGUFValue::INT; v.i = c.get<int16_t>(); break;
    case VT_U32: v.kind = GGUFValue::UINT; v.u = c.get<uint32_t>(); break;
    case VT_I32: v.kind = GGUFValue::INT; v.i = c.get<int32_t>(); break;
    case VT_U64: v.kind = GGUFValue::UINT; v.u = c.get<uint64_t>(); break;
    case VT_I64: v.kind = GGUFValue::INT; v.i = c.get<int64_t>(); break;
    case VT_F32: v.kind = GGUFValue::FLOAT; v.f = c.get<float>(); break;
    case VT_F64: v.kind = GGUFValue::FLOAT; v.f = c.get<double>(); break;
    case VT_BOOL: v.kind = GGUFValue::BOOL; v.u = c.get<uint8_t>() != 0; break;
    case VT_STRING: v.kind = GGUFValue::STRING; v.s = c.str(); break;
    default: throw std::runtime_error("GGUF: unknown metadata value type " + std::to_string(t));
  }
  return v;
}

}  // namespace

GGUFFile::~GGUFFile() { close(); }

void GGUFFile::close() {
  if (map_) munmap(const_cast<uint8_t*>(map_), size_);
  if (fd_ >= 0) ::close(fd_);
  map_ = nullptr;
  fd_ = -1;
  size_ = 0;
  meta_.clear();
  tensors_.clear();
}

void GGUFFile::open(const std::string& path) {
  close();
  fd_ = ::open(path.c_str(), O_RDONLY);
  if (fd_ < 0) throw std::runtime_error("GGUF: cannot open " + path);
  struct stat st;
  if (fstat(fd_, &st) != 0 || st.st_size < 24) {
    close();
    throw std::runtime_error("GGUF: " + path + " is too small to be a GGUF file");
  }
  size_ = (size_t)st.st_size;
  void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
  if (m == MAP_FAILED) {
    close();
    throw std::runtime_error("GGUF: mmap failed for " + path);
  }
  map_ = static_cast<const uint8_t*>(m);
  try {
    Cursor c{map_, size_};
    if (c.get<uint32_t>() != 0x46554747u) throw std::runtime_error("GGUF: bad magic in " + path);
    version_ = c.get<uint32_t>();
    const uint64_t n_tensors = c.get<uint64_t>(), n_kv = c.get<uint64_t>();
    if (n_tensors > (1u << 24) || n_kv > (1u << 24)) throw std::runtime_error("GGUF: unreasonable header counts");
    for (uint64_t k = 0; k < n_kv; ++k) {
      const std::string key = c.str();
      const uint32_t t = c.get<uint32_t>();
      GGUFValue v;
      if (t == VT_ARRAY) {
        v.kind = GGUFValue::ARRAY;
        v.array_type = c.get<uint32_t>();
        v.array_count = c.get<uint64_t>();
        if (v.array_type == VT_STRING) {
          const bool keep = key == "tokenizer.ggml.tokens" || key == "tokenizer.ggml.merges";
          for (uint64_t e = 0; e < v.array_count; ++e) {
            std::string s = c.str();
            if (keep) v.strings.push_back(std::move(s));
          }
        } else {
          const size_t es = scalar_size(v.array_type);
          if (es == 0) throw std::runtime_error("GGUF: array of unsupported element type in key " + key);
          if (v.array_count > size_ / es) throw std::runtime_error("GGUF: array longer than the file in key " + key);
          if (key == "tokenizer.ggml.scores" && v.array_type == VT_F32) {
            v.floats.resize((size_t)v.array_count);
            for (auto& f : v.floats) f = c.get<float>();
          } else if (key == "tokenizer.ggml.token_type" && (v.array_type == VT_U32 || v.array_type == VT_I32)) {
            v.ints.resize((size_t)v.array_count);
            for (auto& t : v.ints) t = c.get<int32_t>();  // (U32 values are reinterpreted, as the reference does)
          } else {
            c.skip((size_t)v.array_count * es);
          }
        }
      } else {
        v = read_scalar(c, t);
      }
      meta_[key] = std::move(v);
    }
    if (const GGUFValue* a = meta("general.alignment")) {
      int64_t al = 0;
      if (a->as_integer(&al) && al > 0) alignment_ = (uint64_t)al;
    }
    tensors_.resize((size_t)n_tensors);
    for (auto& t : tensors_) {
      t.name = c.str();
      const uint32_t nd = c.get<uint32_t>();
      if (nd == 0 || nd > 4) throw std::runtime_error("GGUF: tensor " + t.name + " has " + std::to_string(nd) + " dims");
      uint64_t elems = 1;
      for (uint32_t d = 0; d < nd; ++d) {
        t.dims.push_back(c.get<uint64_t>());
        if (t.dims.back() == 0 || elems > (uint64_t(1) << 40) / t.dims.back())
          throw std::runtime_error("GGUF: tensor " + t.name + " has an unreasonable shape");
        elems *= t.dims.back();
      }
      t.type = c.get<uint32_t>();
      t.offset = c.get<uint64_t>();
      size_t bb = 0, be = 0;
      if (!ggml_block(t.type, &bb, &be))
        throw std::runtime_error("GGUF: tensor " + t.name + " has unsupported ggml type " + std::to_string(t.type));
      if (elems % be != 0)  // gguf_parser.cpp:428-435
        throw std::runtime_error("GGUF: tensor " + t.name + " element count is not a multiple of its block size");
      t.nbytes = (size_t)(elems / be * bb);
    }
    const size_t data_start = (c.pos + alignment_ - 1) / alignment_ * alignment_;
    for (auto& t : tensors_) {
      if (data_start > size_ || t.offset > size_ - data_start || t.nbytes > size_ - data_start - t.offset)
        throw std::runtime_error("GGUF: tensor " + t.name + " extends past the end of the file");
      t.data = map_ + data_start + t.offset;
    }
  } catch (...) {
    close();
    throw;
  }
}

const GGUFTensor* GGUFFile::find(const std::string& name) const {
  for (const auto& t : tensors_)
    if (t.name == name) return &t;
  return nullptr;
}
const GGUFValue* GGUFFile::meta(const std::string& key) const {
  auto it = meta_.find(key);
  return it == meta_.end() ? nullptr : &it->second;
}

}  // namespace tlhost
