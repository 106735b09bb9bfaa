// TEST DOUBLE — not Csound's header.  A small host that offers the subset of Csound 7's C++ plugin
// framework (<plugin.h>: csnd::Plugin, Csound, AuxMem, Vector, Table, AudioSig, plugin<T>(), thread::*)
// which csound/opcode.cpp uses, written from the framework's documented interface so that the opcode
// source can be compiled and driven in tests without a Csound installation (Csound is not in this
// image).  It makes no claim beyond "the opcode logic runs against this API".
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <vector>

typedef double MYFLT;
#ifndef OK
#define OK 0
#endif
#ifndef NOTOK
#define NOTOK (-1)
#endif

struct INSDS {
  uint32_t ksmps;
};

namespace csnd {

struct OpcodeBase {
  INSDS *insdshead = nullptr;
  uint32_t offset = 0, nsmps = 0;
};

class Csound {
 public:
  struct Entry {
    std::string name, outtypes, intypes;
    uint32_t thread;
    std::function<OpcodeBase *()> make;
    std::function<void(OpcodeBase *, Csound *, MYFLT **, MYFLT **)> bind;
    std::function<int(OpcodeBase *)> init, kperf, aperf, deinit;
    std::function<void(OpcodeBase *)> destroy;
  };
  std::map<std::string, Entry> opcodes;
  std::map<int, std::vector<MYFLT>> tables;   // function tables by number
  std::vector<std::string> messages;
  MYFLT dbfs = 1.0;

  int init_error(const std::string &s) { messages.push_back("INIT ERROR: " + s); return NOTOK; }
  int perf_error(const std::string &s, void *) { messages.push_back("PERF ERROR: " + s); return NOTOK; }
  void message(const std::string &s) { messages.push_back(s); }
  MYFLT _0dbfs() { return dbfs; }
};

template <typename T> class AuxMem {
  std::vector<T> v;
 public:
  void allocate(Csound *, int n) { v.assign(n, T()); }
  T &operator[](int n) { return v[n]; }
  T *data() { return v.data(); }
  typename std::vector<T>::iterator begin() { return v.begin(); }
  typename std::vector<T>::iterator end() { return v.end(); }
  uint32_t len() { return (uint32_t)v.size(); }
};

template <typename T> class Vector {
  std::vector<T> v;
 public:
  void init(Csound *, int size, void * = nullptr) { v.assign(size, T()); }
  uint32_t len() { return (uint32_t)v.size(); }
  T &operator[](int n) { return v[n]; }
  T *data_array() { return v.data(); }
};

class Table {
  std::vector<MYFLT> *t = nullptr;
 public:
  int init(Csound *cs, MYFLT *arg) {
    auto it = cs->tables.find((int)*arg);
    if (it == cs->tables.end()) return NOTOK;
    t = &it->second;
    return OK;
  }
  uint32_t len() { return t ? (uint32_t)t->size() : 0; }
  MYFLT &operator[](int n) { return (*t)[n]; }
};

class AudioSig {
  MYFLT *sig;
 public:
  AudioSig(OpcodeBase *, MYFLT *s, bool = false) : sig(s) {}
  MYFLT &operator[](int n) { return sig[n]; }
};

template <std::size_t N> class Param {
 public:
  MYFLT *ptrs[N];
  MYFLT &operator[](int n) { return *ptrs[n]; }
  MYFLT *operator()(int n) { return ptrs[n]; }
  template <typename T> Vector<T> &vector_data(int n) { return *reinterpret_cast<Vector<T> *>(ptrs[n]); }
};

template <std::size_t N, std::size_t M> struct Plugin : OpcodeBase {
  Param<N> outargs;
  Param<M> inargs;
  Csound *csound = nullptr;
  int init() { return OK; }
  int kperf() { return OK; }
  int aperf() { return OK; }
  int deinit() { return OK; }
};

namespace thread {
enum { i = 1, k = 2, ik = 3, a = 4, ia = 5 };
}

template <typename T>
int plugin(Csound *cs, const char *name, const char *oargs, const char *iargs, uint32_t thr, uint32_t = 0) {
  Csound::Entry e;
  e.name = name;
  e.outtypes = oargs;
  e.intypes = iargs;
  e.thread = thr;
  e.make = []() -> OpcodeBase * { return new T(); };
  e.bind = [](OpcodeBase *b, Csound *c, MYFLT **outs, MYFLT **ins) {
    T *p = static_cast<T *>(b);
    p->csound = c;
    for (std::size_t k = 0; k < sizeof(p->outargs.ptrs) / sizeof(MYFLT *); k++) p->outargs.ptrs[k] = outs[k];
    for (std::size_t k = 0; k < sizeof(p->inargs.ptrs) / sizeof(MYFLT *); k++) p->inargs.ptrs[k] = ins[k];
  };
  e.init = [](OpcodeBase *b) { return static_cast<T *>(b)->init(); };
  e.kperf = [](OpcodeBase *b) { return static_cast<T *>(b)->kperf(); };
  e.aperf = [](OpcodeBase *b) { return static_cast<T *>(b)->aperf(); };
  e.deinit = [](OpcodeBase *b) { return static_cast<T *>(b)->deinit(); };
  e.destroy = [](OpcodeBase *b) { delete static_cast<T *>(b); };
  cs->opcodes[name] = e;
  return OK;
}

void on_load(Csound *);

}  // namespace csnd
