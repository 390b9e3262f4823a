// Launch programs (include/dana_hip.h, "launch programs"): a recorded list of C-ABI calls of THIS library, event records and
// event waits, re-issued from one C loop -- dana_amd/program.py records the eager step's launch sequence once and replays
// it without the Python between the launches (train.py:125-143's iteration: ~1 500 calls, 11-13 ms of host time eagerly).
// Host-only code: no kernels here. The entry points it re-issues have plain C signatures (ints, longs, floats, doubles,
// pointers), so one generic caller serves them all: libffi, the same library Python's ctypes binding of this ABI calls
// through, loaded at run time (dlopen: there is no libffi header in the image, so the few declarations needed are below).
#include "common.h"
#include "../../include/dana_hip.h"
#include <dlfcn.h>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

// ---- libffi's public ABI (libffi 3.3 / 3.4, x86-64 System V), as much of it as this file uses -----------------------------
struct ffi_type_ {
  size_t size;
  unsigned short alignment;
  unsigned short type;
  ffi_type_** elements;
};
struct ffi_cif_ {
  int abi;
  unsigned nargs;
  ffi_type_** arg_types;
  ffi_type_* rtype;
  unsigned bytes;
  unsigned flags;
  char reserve[64];  // (room for a build of libffi that appends target-specific fields)
};
constexpr int FFI_UNIX64_ = 2;  // enum ffi_abi { FFI_FIRST_ABI = 1, FFI_UNIX64, ... } on x86-64 Linux
using ffi_prep_cif_t = int (*)(ffi_cif_*, int, unsigned, ffi_type_*, ffi_type_**);
using ffi_call_t = void (*)(ffi_cif_*, void (*)(void), void*, void**);

struct Ffi {
  void* handle = nullptr;
  ffi_prep_cif_t prep = nullptr;
  ffi_call_t call = nullptr;
  ffi_type_ *t_sint32 = nullptr, *t_sint64 = nullptr, *t_float = nullptr, *t_double = nullptr, *t_pointer = nullptr;
  bool ok = false;
  Ffi() {
#if !defined(__x86_64__)
    return;  // the ffi_cif layout above is the x86-64 one
#endif
    for (const char* name : {"libffi.so.8", "libffi.so.7", "libffi.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) return;
    prep = (ffi_prep_cif_t)dlsym(handle, "ffi_prep_cif");
    call = (ffi_call_t)dlsym(handle, "ffi_call");
    t_sint32 = (ffi_type_*)dlsym(handle, "ffi_type_sint32");
    t_sint64 = (ffi_type_*)dlsym(handle, "ffi_type_sint64");
    t_float = (ffi_type_*)dlsym(handle, "ffi_type_float");
    t_double = (ffi_type_*)dlsym(handle, "ffi_type_double");
    t_pointer = (ffi_type_*)dlsym(handle, "ffi_type_pointer");
    ok = prep && call && t_sint32 && t_sint64 && t_float && t_double && t_pointer;
  }
};
Ffi& ffi() {
  static Ffi f;
  return f;
}

struct Signature {
  std::vector<ffi_type_*> types;
  ffi_cif_ cif;
};

enum Kind : int { K_CALL = 0, K_RECORD = 1, K_WAIT = 2 };

struct Entry {
  int kind;
  void* fn;            // K_CALL: the entry point; K_RECORD / K_WAIT: the hipEvent_t
  void* stream;        // K_RECORD / K_WAIT
  Signature* sig;      // K_CALL
  unsigned first_arg;  // K_CALL: index of its first word in Program::words
  unsigned nargs;
};

struct Program {
  std::vector<Entry> entries;
  std::vector<unsigned long long> words;                   // every call's arguments as 64-bit words
  std::vector<void*> argp;                                 // scratch: ffi_call's array of pointers to them
  std::map<std::string, std::unique_ptr<Signature>> sigs;  // one prepared call interface per distinct signature
  size_t max_args = 0;
};

}  // namespace

extern "C" {

int dana_program_create(void** program_out) {
  DANA_CHECK_ARG(program_out, "dana_program_create: null pointer");
  if (!ffi().ok) {
    dana_set_error("dana_program_create: libffi could not be loaded (dlopen libffi.so.8 / .7): replay from Python instead");
    return DANA_ERR_ARG;
  }
  *program_out = new Program();
  return DANA_OK;
}

int dana_program_destroy(void* program) {
  delete (Program*)program;
  return DANA_OK;
}

int dana_program_add_call(void* program, void* entry_point, const char* signature, const unsigned long long* words, int nargs) {
  Program* p = (Program*)program;
  DANA_CHECK_ARG(p && entry_point && signature && nargs >= 0 && nargs <= 64 && (nargs == 0 || words) &&
                     (int)strlen(signature) == nargs,
                 "dana_program_add_call: bad arguments");
  auto it = p->sigs.find(signature);
  if (it == p->sigs.end()) {
    std::unique_ptr<Signature> s(new Signature());
    Ffi& f = ffi();
    for (int i = 0; i < nargs; ++i) {
      switch (signature[i]) {
        case 'i': s->types.push_back(f.t_sint32); break;
        case 'l': s->types.push_back(f.t_sint64); break;  // long / size_t / unsigned long long
        case 'p': s->types.push_back(f.t_pointer); break;
        case 'f': s->types.push_back(f.t_float); break;
        case 'd': s->types.push_back(f.t_double); break;
        default: DANA_CHECK_ARG(false, "dana_program_add_call: signature character '%c' (one of i l p f d)", signature[i]);
      }
    }
    const int rc = f.prep(&s->cif, FFI_UNIX64_, (unsigned)nargs, f.t_sint32, s->types.data());
    DANA_CHECK_ARG(rc == 0, "dana_program_add_call: ffi_prep_cif failed (%d)", rc);
    it = p->sigs.emplace(signature, std::move(s)).first;
  }
  Entry e{K_CALL, entry_point, nullptr, it->second.get(), (unsigned)p->words.size(), (unsigned)nargs};
  p->words.insert(p->words.end(), words, words + nargs);
  p->entries.push_back(e);
  if ((size_t)nargs > p->max_args) p->max_args = nargs;
  return DANA_OK;
}

int dana_program_add_event_record(void* program, void* event, dana_stream_t stream) {
  Program* p = (Program*)program;
  DANA_CHECK_ARG(p && event, "dana_program_add_event_record: null pointer");
  p->entries.push_back(Entry{K_RECORD, event, stream, nullptr, 0, 0});
  return DANA_OK;
}

int dana_program_add_event_wait(void* program, void* event, dana_stream_t stream) {
  Program* p = (Program*)program;
  DANA_CHECK_ARG(p && event, "dana_program_add_event_wait: null pointer");
  p->entries.push_back(Entry{K_WAIT, event, stream, nullptr, 0, 0});
  return DANA_OK;
}

// The three torch ops a recorded backward consists of besides this library's launches (zeros / zero_, copy_ / clone between
// device tensors, grad.add_): as entry points, so that a launch program re-issues them from the same C loop
int dana_fill_zero(void* dst, size_t bytes, dana_stream_t stream) {
  if (bytes == 0) return DANA_OK;
  DANA_CHECK_ARG(dst, "dana_fill_zero: null pointer");
  const hipError_t e = hipMemsetAsync(dst, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) {
    dana_set_error("dana_fill_zero: %s", hipGetErrorString(e));
    return DANA_ERR_HIP;
  }
  return DANA_OK;
}

int dana_copy_d2d(void* dst, const void* src, size_t bytes, dana_stream_t stream) {
  if (bytes == 0) return DANA_OK;
  DANA_CHECK_ARG(dst && src, "dana_copy_d2d: null pointer");
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    dana_set_error("dana_copy_d2d: %s", hipGetErrorString(e));
    return DANA_ERR_HIP;
  }
  return DANA_OK;
}

int dana_program_size(void* program) {
  return program ? (int)((Program*)program)->entries.size() : 0;
}

// Re-issue entries [begin, end) in order. Stops at the first failing entry (its own error text stays in dana_last_error()).
int dana_program_run(void* program, int begin, int end) {
  Program* p = (Program*)program;
  DANA_CHECK_ARG(p && begin >= 0 && end >= begin && end <= (int)p->entries.size(), "dana_program_run: bad range [%d, %d)", begin, end);
  Ffi& f = ffi();
  if (p->argp.size() < p->max_args) p->argp.resize(p->max_args);
  void** argp = p->argp.data();
  for (int i = begin; i < end; ++i) {
    const Entry& e = p->entries[i];
    if (e.kind == K_CALL) {
      unsigned long long* w = p->words.data() + e.first_arg;
      for (unsigned a = 0; a < e.nargs; ++a) argp[a] = w + a;  // little-endian: an int / float is the word's low half
      long rc = 0;                                              // (libffi widens integral results to a full register)
      f.call(&e.sig->cif, (void (*)(void))e.fn, &rc, argp);
      if ((int)rc != 0) return (int)rc;
    } else if (e.kind == K_RECORD) {
      const hipError_t err = hipEventRecord((hipEvent_t)e.fn, (hipStream_t)e.stream);
      if (err != hipSuccess) {
        dana_set_error("dana_program_run: hipEventRecord (entry %d): %s", i, hipGetErrorString(err));
        return DANA_ERR_HIP;
      }
    } else {
      const hipError_t err = hipStreamWaitEvent((hipStream_t)e.stream, (hipEvent_t)e.fn, 0);
      if (err != hipSuccess) {
        dana_set_error("dana_program_run: hipStreamWaitEvent (entry %d): %s", i, hipGetErrorString(err));
        return DANA_ERR_HIP;
      }
    }
  }
  return DANA_OK;
}

}  // extern "C"
