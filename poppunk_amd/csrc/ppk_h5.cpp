// Bulk reader of PopPUNK sketch databases (`<db>/<db>.h5`, layout PopPUNK/web.py:14-61; the
// reference reads them sample by sample through h5py, PopPUNK/sketchlib.py:86-88,124-133, and
// pp-sketchlib through HighFive [EXT]).  Host code only.
//
// Two backends behind one handle:
//   1 "walker"  -- the file is mmap-ed and its structures are read directly: superblock 0/1, version-1
//                  object headers, symbol-table groups (B-tree v1 + local heap), contiguous or compact
//                  little-endian datasets, attribute messages v1-3.  That is what h5py and HighFive write
//                  with their default (earliest-format) settings, i.e. every PopPUNK database.  Samples
//                  are independent, so the per-sample work runs on several threads; nothing is decoded
//                  that is not needed (no property lists, no type conversion paths, no metadata cache).
//                  (A /sketches group that libhdf5 has converted to new-style links -- one non-ASCII sample name does
//                  that -- is LISTED through libhdf5, one H5Literate, and its samples read here all the same.)
//   2 "libhdf5" -- anything the walker does not recognise (superblock 2/3, new-style groups elsewhere,
//                  chunked / filtered / big-endian / shared-message objects) is read through libhdf5's own
//                  C API, dlopen-ed at first need: one H5Fopen, per sample one H5Gopen2 and nk
//                  H5Dopen2 + H5Dread straight into the caller's array.  ~23 us per dataset against the
//                  walker's < 1 us; still an order of magnitude below a per-dataset Python loop.
// Every offset the walker follows is bounds-checked against the mapping: a truncated or foreign file
// makes it decline (-> backend 2, whose errors are libhdf5's), never read outside the file.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ppk.h"

void ppk_set_error(const std::string &msg);
int ppk_fail(int code, const std::string &msg);

namespace {

constexpr uint64_t UNDEF = ~0ull;

struct Unsupported {
  std::string why;
};

// ---- the walker --------------------------------------------------------------------------------------

struct Map {
  const uint8_t *p = nullptr;
  size_t size = 0;
  uint64_t base = 0;
  const uint8_t *at(uint64_t off, uint64_t len) const {
    if (off == UNDEF || off + base < off || off + base > size || len > size - (off + base))
      throw Unsupported{"address outside the file"};
    return p + base + off;
  }
  uint64_t u(uint64_t off, int bytes) const {
    const uint8_t *q = at(off, bytes);
    uint64_t v = 0;
    memcpy(&v, q, bytes);      // little-endian host (x86-64)
    return v;
  }
};

static inline uint64_t rd(const uint8_t *q, int bytes) {
  uint64_t v = 0;
  memcpy(&v, q, bytes);
  return v;
}

struct Msg {
  uint16_t type;
  uint8_t flags;
  const uint8_t *data;
  size_t size;
};

// All messages of a version-1 object header, continuation blocks included.
static void object_messages(const Map &m, uint64_t addr, std::vector<Msg> &out) {
  out.clear();
  const uint8_t *h = m.at(addr, 16);
  if (h[0] != 1) throw Unsupported{"object header version " + std::to_string(h[0])};
  size_t nmesg = rd(h + 2, 2);
  uint64_t hsize = rd(h + 8, 4);
  struct Block {
    uint64_t off, len;
  };
  std::vector<Block> blocks{{addr + 16, hsize}};
  for (size_t b = 0; b < blocks.size() && out.size() < nmesg; b++) {
    if (blocks.size() > 64) throw Unsupported{"object header with too many continuation blocks"};
    const uint8_t *q = m.at(blocks[b].off, blocks[b].len);
    uint64_t pos = 0, len = blocks[b].len;
    while (pos + 8 <= len && out.size() < nmesg) {
      uint16_t type = (uint16_t)rd(q + pos, 2);
      size_t sz = rd(q + pos + 2, 2);
      uint8_t flags = q[pos + 4];
      if (pos + 8 + sz > len) throw Unsupported{"object header message overruns its block"};
      const uint8_t *d = q + pos + 8;
      out.push_back({type, flags, d, sz});
      if (type == 0x0010) {        // continuation
        if (sz < 16) throw Unsupported{"short continuation message"};
        blocks.push_back({rd(d, 8), rd(d + 8, 8)});
      }
      pos += 8 + sz;
    }
  }
}

struct Link {
  std::string name;
  uint64_t ohdr;
  uint32_t cache;
  uint64_t btree, heap;      // valid when cache == 1
};

// Names and object headers of a symbol-table group, in B-tree (= name) order.
static void group_links(const Map &m, uint64_t btree, uint64_t heap, std::vector<Link> &out, size_t limit) {
  const uint8_t *hp = m.at(heap, 32);
  if (memcmp(hp, "HEAP", 4) != 0) throw Unsupported{"local heap signature"};
  uint64_t hsz = rd(hp + 8, 8), hdata = rd(hp + 24, 8);
  const uint8_t *names = m.at(hdata, hsz);
  // depth-first, children left to right
  // A crafted file may point every entry of an internal node at the same child: levels must strictly descend
  // (a child of a level-L node is a level-(L-1) node) and the walk visits at most as many nodes as the file
  // could hold (a node is at least 24 bytes), so it always ends.
  struct Frame {
    uint64_t addr;
    size_t next;
    int want_level;      // -1: the root, any level
  };
  std::vector<Frame> frames{{btree, 0, -1}};
  size_t visited = 0;
  const size_t max_nodes = limit + 64;
  while (!frames.empty()) {
    if (frames.size() > 32) throw Unsupported{"B-tree too deep"};
    Frame &f = frames.back();
    const uint8_t *n = m.at(f.addr, 24);
    if (memcmp(n, "TREE", 4) != 0 || n[4] != 0) throw Unsupported{"group B-tree node"};
    unsigned level = n[5];
    if (f.want_level >= 0 && (int)level != f.want_level) throw Unsupported{"B-tree levels do not descend"};
    if (f.next == 0 && ++visited > max_nodes) throw Unsupported{"B-tree larger than the file can hold"};
    size_t used = rd(n + 6, 2);
    const uint8_t *body = m.at(f.addr + 24, (2 * used + 1) * 8);
    if (f.next >= used) {
      frames.pop_back();
      continue;
    }
    uint64_t child = rd(body + 8 + 16 * f.next, 8);
    f.next++;
    if (level > 0) {
      frames.push_back({child, 0, (int)level - 1});
      continue;
    }
    if (++visited > max_nodes) throw Unsupported{"B-tree larger than the file can hold"};
    const uint8_t *s = m.at(child, 8);
    if (memcmp(s, "SNOD", 4) != 0) throw Unsupported{"symbol table node"};
    size_t nsym = rd(s + 6, 2);
    const uint8_t *e = m.at(child + 8, nsym * 40);
    for (size_t i = 0; i < nsym; i++, e += 40) {
      uint64_t noff = rd(e, 8);
      if (noff >= hsz) throw Unsupported{"link name outside the heap"};
      const void *z = memchr(names + noff, 0, hsz - noff);
      if (!z) throw Unsupported{"unterminated link name"};
      Link l;
      l.name.assign((const char *)names + noff, (const char *)z);
      l.ohdr = rd(e + 8, 8);
      l.cache = (uint32_t)rd(e + 16, 4);
      l.btree = l.cache == 1 ? rd(e + 24, 8) : UNDEF;
      l.heap = l.cache == 1 ? rd(e + 32, 8) : UNDEF;
      if (l.cache == 2) throw Unsupported{"soft link"};
      out.push_back(std::move(l));
      if (out.size() > limit) throw Unsupported{"group larger than the file can hold"};
    }
  }
}

// B-tree and heap of a group, ALWAYS from its object header.  A symbol table entry also caches the two addresses of
// the group it points to, but that cache goes stale: when a link that a symbol table cannot hold is added to an
// old-style group -- a sample whose name is not plain ASCII is enough -- the library converts the group to link
// messages, drops the Symbol Table message from its header and leaves the old B-tree, and the parent's cached
// addresses, behind.  Read through the cache such a database shows the samples it had BEFORE the conversion and
// nothing else (tests/test_h5bulk.py::test_odd_but_legal_databases found exactly that).  A header without the
// message is a new-style group: the direct reader declines and libhdf5 reads the file.
static void group_of_object(const Map &m, uint64_t ohdr, uint64_t &btree, uint64_t &heap, std::vector<Msg> &scratch) {
  object_messages(m, ohdr, scratch);
  for (const Msg &g : scratch)
    if (g.type == 0x0011) {
      if (g.flags & 2 || g.size < 16) throw Unsupported{"symbol table message"};
      btree = rd(g.data, 8);
      heap = rd(g.data + 8, 8);
      return;
    }
  throw Unsupported{"group without a symbol table (new-style links)"};
}

struct TypeInfo {
  int cls;          // 0 integer, 1 float
  size_t size;
  bool is_signed;
  size_t msg_bytes;   // bytes the datatype message occupies
};

static TypeInfo parse_type(const uint8_t *d, size_t avail) {
  if (avail < 8) throw Unsupported{"short datatype message"};
  TypeInfo t;
  t.cls = d[0] & 0x0f;
  int ver = d[0] >> 4;
  if (ver < 1 || ver > 3) throw Unsupported{"datatype version"};
  t.size = rd(d + 4, 4);
  if (t.cls == 0) {
    if (d[1] & 1) throw Unsupported{"big-endian integers"};
    t.is_signed = (d[1] & 8) != 0;
    t.msg_bytes = 8 + 4;
    if (t.size != 1 && t.size != 2 && t.size != 4 && t.size != 8) throw Unsupported{"integer size"};
    if (avail < 12 || rd(d + 8, 2) != 0 || rd(d + 10, 2) != 8 * t.size) throw Unsupported{"integer with padding bits"};
  } else if (t.cls == 1) {
    if ((d[1] & 1) || (d[1] & 0x40)) throw Unsupported{"non-little-endian floats"};
    t.is_signed = true;
    t.msg_bytes = 8 + 12;
    if (avail < 20) throw Unsupported{"short float datatype"};
    // IEEE layouts only: (sign 31, exp 23/8, mant 0/23) or (63, 52/11, 0/52)
    unsigned epos = d[12], esz = d[13], mpos = d[14], msz = d[15];
    bool f32 = t.size == 4 && epos == 23 && esz == 8 && mpos == 0 && msz == 23;
    bool f64 = t.size == 8 && epos == 52 && esz == 11 && mpos == 0 && msz == 52;
    if (!f32 && !f64) throw Unsupported{"non-IEEE float"};
  } else {
    throw Unsupported{"datatype class " + std::to_string(t.cls)};
  }
  return t;
}

// number of elements of a dataspace message; bytes it occupies in *msg_bytes
static uint64_t parse_space(const uint8_t *d, size_t avail, size_t *msg_bytes) {
  if (avail < 4) throw Unsupported{"short dataspace message"};
  int ver = d[0], rank = d[1], flags = d[2];
  size_t head;
  if (ver == 1)
    head = 8;
  else if (ver == 2) {
    head = 4;
    if (d[3] == 2) throw Unsupported{"null dataspace"};
  } else
    throw Unsupported{"dataspace version"};
  size_t need = head + (size_t)rank * 8 * ((flags & 1) ? 2 : 1);
  if (avail < need) throw Unsupported{"short dataspace message"};
  uint64_t n = 1;
  for (int i = 0; i < rank; i++) {
    uint64_t dim = rd(d + head + 8 * i, 8);
    if (dim && n > (1ull << 62) / dim) throw Unsupported{"dataspace too large"};
    n *= dim;
  }
  if (msg_bytes) *msg_bytes = need;
  return n;
}

struct Attr {
  std::string name;
  TypeInfo type;
  uint64_t count;
  const uint8_t *data;
};

static bool parse_attr(const Msg &g, Attr &a) {
  if (g.flags & 2) throw Unsupported{"shared attribute"};
  const uint8_t *d = g.data;
  if (g.size < 8) throw Unsupported{"short attribute message"};
  int ver = d[0];
  if (ver < 1 || ver > 3) throw Unsupported{"attribute version"};
  if (ver >= 2 && (d[1] & 3)) throw Unsupported{"attribute with shared type or space"};
  size_t nsz = rd(d + 2, 2), tsz = rd(d + 4, 2), ssz = rd(d + 6, 2);
  size_t pos = ver == 3 ? 9 : 8;
  auto pad = [&](size_t x) { return ver == 1 ? (x + 7) & ~(size_t)7 : x; };
  if (pos + pad(nsz) + pad(tsz) + pad(ssz) > g.size || nsz == 0) throw Unsupported{"attribute message layout"};
  a.name.assign((const char *)d + pos, strnlen((const char *)d + pos, nsz));
  pos += pad(nsz);
  if (tsz < 8) throw Unsupported{"short datatype message"};
  int cls = d[pos] & 0x0f;
  if (cls == 8) {                                   // enum (h5py's bool): read as its base integer
    size_t esz = rd(d + pos + 4, 4);
    if (esz != 1 && esz != 2 && esz != 4 && esz != 8) return false;
    a.type = TypeInfo{0, esz, false, 0};
  } else if (cls != 0 && cls != 1) {
    return false;                                   // strings ...: not one this reader needs
  } else
    a.type = parse_type(d + pos, tsz);
  pos += pad(tsz);
  a.count = parse_space(d + pos, ssz, nullptr);
  pos += pad(ssz);
  if (a.count > (g.size - pos) / (a.type.size ? a.type.size : 1)) throw Unsupported{"attribute data overruns its message"};
  a.data = d + pos;
  return true;
}

static int64_t attr_int(const Attr &a, uint64_t i) {
  const uint8_t *q = a.data + i * a.type.size;
  if (a.type.cls == 1) return (int64_t)(a.type.size == 4 ? (double)*(const float *)q : *(const double *)q);
  uint64_t v = rd(q, (int)a.type.size);
  if (a.type.is_signed && a.type.size < 8 && (v >> (8 * a.type.size - 1))) v |= ~0ull << (8 * a.type.size);
  return (int64_t)v;
}

static double attr_double(const Attr &a, uint64_t i) {
  const uint8_t *q = a.data + i * a.type.size;
  if (a.type.cls == 1) {
    if (a.type.size == 4) {
      float f;
      memcpy(&f, q, 4);
      return f;
    }
    double f;
    memcpy(&f, q, 8);
    return f;
  }
  return (double)attr_int(a, i);
}

struct Extent {
  const uint8_t *data;    // nullptr: storage never allocated (all zero)
  uint64_t bytes;
};

// A dataset of unsigned/signed 64-bit little-endian integers with `words` elements: where its bytes are.
static Extent dataset_extent(const Map &m, uint64_t ohdr, uint64_t *count, std::vector<Msg> &scratch) {
  object_messages(m, ohdr, scratch);
  const Msg *space = nullptr, *type = nullptr, *layout = nullptr;
  for (const Msg &g : scratch) {
    if (g.type == 0x0001) space = &g;
    if (g.type == 0x0003) type = &g;
    if (g.type == 0x0008) layout = &g;
    if (g.type == 0x000B) throw Unsupported{"filtered dataset"};
    if (g.type == 0x0007) throw Unsupported{"external storage"};
  }
  if (!space || !type || !layout) throw Unsupported{"dataset header lacks space, type or layout"};
  if ((space->flags | type->flags | layout->flags) & 2) throw Unsupported{"shared header message"};
  TypeInfo t = parse_type(type->data, type->size);
  if (t.cls != 0 || t.size != 8) throw Unsupported{"sketch dataset is not a 64-bit integer"};
  *count = parse_space(space->data, space->size, nullptr);
  const uint8_t *l = layout->data;
  if (layout->size < 2 || l[0] != 3) throw Unsupported{"data layout message version"};
  if (l[1] == 1) {
    if (layout->size < 18) throw Unsupported{"short layout message"};
    uint64_t addr = rd(l + 2, 8), bytes = rd(l + 10, 8);
    if (bytes != *count * 8) throw Unsupported{"dataset size does not match its space"};
    // (storage never allocated: the values are the dataset's fill value, which the library knows how to find)
    if (addr == UNDEF) throw Unsupported{"dataset without allocated storage"};
    return {m.at(addr, bytes), bytes};
  }
  if (l[1] == 0) {
    if (layout->size < 4) throw Unsupported{"short layout message"};
    uint64_t bytes = rd(l + 2, 2);
    if (bytes != *count * 8 || 4 + bytes > layout->size) throw Unsupported{"compact dataset size"};
    return {l + 4, bytes};
  }
  throw Unsupported{"chunked dataset"};
}

// ---- libhdf5 through dlopen ---------------------------------------------------------------------------

typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

struct Hdf5 {
  void *so = nullptr;
  std::string path;
  herr_t (*H5open)();
  herr_t (*H5Eset_auto2)(hid_t, void *, void *);
  herr_t (*H5get_libversion)(unsigned *, unsigned *, unsigned *);
  hid_t (*H5Fopen)(const char *, unsigned, hid_t);
  herr_t (*H5Fclose)(hid_t);
  hid_t (*H5Gopen2)(hid_t, const char *, hid_t);
  herr_t (*H5Gclose)(hid_t);
  hid_t (*H5Dopen2)(hid_t, const char *, hid_t);
  herr_t (*H5Dclose)(hid_t);
  hid_t (*H5Dget_space)(hid_t);
  herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *);
  long long (*H5Sget_simple_extent_npoints)(hid_t);
  herr_t (*H5Sclose)(hid_t);
  int (*H5Aexists)(hid_t, const char *);
  hid_t (*H5Aopen)(hid_t, const char *, hid_t);
  hid_t (*H5Aget_space)(hid_t);
  herr_t (*H5Aread)(hid_t, hid_t, void *);
  herr_t (*H5Aclose)(hid_t);
  hid_t (*H5Aget_type)(hid_t);
  size_t (*H5Tget_size)(hid_t);
  herr_t (*H5Tclose)(hid_t);
  int (*H5Lexists)(hid_t, const char *, hid_t);
  herr_t (*H5Literate)(hid_t, int, int, hsize_t *, herr_t (*)(hid_t, const char *, const void *, void *), void *);
  hid_t t_u64, t_i64, t_f64;
};

static std::mutex g_h5_mutex;          // libhdf5 is used from one thread at a time
static Hdf5 *g_h5 = nullptr;
static std::string g_h5_hint;

static Hdf5 *hdf5_library(std::string &err) {
  if (g_h5) return g_h5;
  std::vector<std::string> cands;
  if (!g_h5_hint.empty()) cands.push_back(g_h5_hint);
  if (const char *e = getenv("HDF5_LIB"))
    if (*e) cands.push_back(e);
  for (const char *c : {"libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
                        "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"})
    cands.push_back(c);
  std::string tried;
  for (const std::string &c : cands) {
    void *so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!so) {
      tried += c + "; ";
      continue;
    }
    Hdf5 *h = new Hdf5;
    h->so = so;
    h->path = c;
    bool ok = true;
    auto sym = [&](const char *name, bool required = true) -> void * {
      void *p = dlsym(so, name);
      if (!p && required) ok = false;
      return p;
    };
#define PPK_H5SYM(f) *(void **)(&h->f) = sym(#f)
    PPK_H5SYM(H5open);
    PPK_H5SYM(H5Eset_auto2);
    PPK_H5SYM(H5get_libversion);
    PPK_H5SYM(H5Fopen);
    PPK_H5SYM(H5Fclose);
    PPK_H5SYM(H5Gopen2);
    PPK_H5SYM(H5Gclose);
    PPK_H5SYM(H5Dopen2);
    PPK_H5SYM(H5Dclose);
    PPK_H5SYM(H5Dget_space);
    PPK_H5SYM(H5Dread);
    PPK_H5SYM(H5Sget_simple_extent_npoints);
    PPK_H5SYM(H5Sclose);
    PPK_H5SYM(H5Aexists);
    PPK_H5SYM(H5Aopen);
    PPK_H5SYM(H5Aget_space);
    PPK_H5SYM(H5Aread);
    PPK_H5SYM(H5Aclose);
    PPK_H5SYM(H5Aget_type);
    PPK_H5SYM(H5Tget_size);
    PPK_H5SYM(H5Tclose);
    PPK_H5SYM(H5Lexists);
#undef PPK_H5SYM
    // the callback's `info` argument differs between 1.10 and 1.12+; only the name is used
    void *it = sym("H5Literate", false);
    if (!it) it = sym("H5Literate1", false);
    if (!it) it = sym("H5Literate2", false);
    if (!it) ok = false;
    *(void **)(&h->H5Literate) = it;
    unsigned maj = 0, mnr = 0, rel = 0;
    if (ok) h->H5get_libversion(&maj, &mnr, &rel);
    if (!ok || maj != 1 || mnr < 10) {      // 64-bit hid_t from 1.10
      tried += c + " (unusable); ";
      delete h;
      dlclose(so);
      continue;
    }
    h->H5open();
    h->H5Eset_auto2(0, nullptr, nullptr);
    void *a = dlsym(so, "H5T_NATIVE_UINT64_g"), *b = dlsym(so, "H5T_NATIVE_INT64_g"), *d = dlsym(so, "H5T_NATIVE_DOUBLE_g");
    if (!a || !b || !d) {
      tried += c + " (no native type ids); ";
      delete h;
      continue;
    }
    h->t_u64 = *(hid_t *)a;
    h->t_i64 = *(hid_t *)b;
    h->t_f64 = *(hid_t *)d;
    g_h5 = h;
    return h;
  }
  err = "libhdf5 (>= 1.10) not found; set HDF5_LIB=/path/to/libhdf5.so (tried: " + tried + ")";
  return nullptr;
}

}  // namespace

// ---- the handle -------------------------------------------------------------------------------------

struct ppk_h5 {
  int backend = 0;       // 1 walker, 2 libhdf5
  bool forced = false;   // the caller asked for this backend: no hand-over to the other
  std::string path;
  std::string declined;  // why the walker passed the file on (backend 2)
  // walker
  int fd = -1;
  Map map;
  std::vector<Link> samples;                       // /sketches, name order
  std::unordered_map<std::string, size_t> index;   // name -> samples[]
  bool has_random = false;
  // libhdf5
  Hdf5 *lib = nullptr;
  hid_t file = -1, top = -1;
  std::vector<std::string> names5;
  bool names5_read = false;
  // parameters of the first sample
  size_t s64 = 0, bbits = 0;
  std::vector<int64_t> kmers;
  bool listed_by_library = false;   // /sketches is a new-style group: names and addresses came from H5Literate
  int codon_phased = -1;      // attribute of /sketches: -1 absent
  uint64_t sketches_ohdr = UNDEF;
};

namespace {

static void sample_params_of(const ppk_h5 *h, size_t idx, size_t *s64, size_t *bbits, std::vector<int64_t> *kmers) {
  std::vector<Msg> msgs;
  object_messages(h->map, h->samples[idx].ohdr, msgs);
  *s64 = *bbits = 0;
  kmers->clear();
  for (const Msg &g : msgs) {
    if (g.type != 0x000C) continue;
    Attr a;
    if (!parse_attr(g, a) || a.count == 0) continue;
    if (a.name == "sketchsize64") *s64 = (size_t)attr_int(a, 0);
    if (a.name == "bbits") *bbits = (size_t)attr_int(a, 0);
    if (a.name == "kmers") {
      kmers->resize(a.count);
      for (uint64_t i = 0; i < a.count; i++) (*kmers)[i] = attr_int(a, i);
    }
  }
}

static void walker_sample_params(ppk_h5 *h) {
  if (h->samples.empty()) return;
  sample_params_of(h, 0, &h->s64, &h->bbits, &h->kmers);
  if (h->s64 == 0 || h->bbits == 0) throw Unsupported{"first sample lacks sketchsize64 / bbits attributes"};
  if (h->s64 > h->map.size || h->bbits > 64 || h->s64 * h->bbits * 8 > h->map.size)
    throw Unsupported{"sketch size attributes larger than the file"};
}

// /sketches as a NEW-style group inside an otherwise old-style file (what one non-ASCII sample name leaves behind: the
// links live in messages or in a fractal heap indexed by a v2 B-tree).  The direct reader does not decode those; it
// asks libhdf5 for the one thing it needs from them -- every link's name and object address, one H5Literate over the
// group (the 1.10 / H5Literate1 form of the callback's `info`, whose hard-link address sits at byte 24) -- and reads
// the samples, whose own groups are ordinary symbol tables, itself.  An address that does not lead to such a group
// makes the read decline like any other surprise, so a wrong guess about `info` costs speed, never correctness.
struct HybridList {
  std::vector<Link> *out;
};
static herr_t hybrid_collect(hid_t, const char *name, const void *info, void *ud) {
  const uint8_t *p = static_cast<const uint8_t *>(info);
  int type;
  memcpy(&type, p, 4);
  if (type != 0) return 0;                 // (hard links only: H5L_TYPE_HARD)
  Link l;
  l.name = name;
  memcpy(&l.ohdr, p + 24, 8);
  l.cache = 0;
  l.btree = l.heap = UNDEF;
  static_cast<HybridList *>(ud)->out->push_back(std::move(l));
  return 0;
}
static bool hybrid_list_sketches(ppk_h5 *h) {
  std::lock_guard<std::mutex> g(g_h5_mutex);
  std::string err;
  Hdf5 *L = hdf5_library(err);
  if (!L) return false;
  const hid_t file = L->H5Fopen(h->path.c_str(), 0, 0);
  if (file < 0) return false;
  const hid_t top = L->H5Gopen2(file, "sketches", 0);
  bool ok = false;
  if (top >= 0) {
    HybridList hl{&h->samples};
    hsize_t idx = 0;
    ok = L->H5Literate(top, 0 /* H5_INDEX_NAME */, 0 /* H5_ITER_INC */, &idx, hybrid_collect, &hl) >= 0;
    L->H5Gclose(top);
  }
  L->H5Fclose(file);
  if (!ok) h->samples.clear();
  std::sort(h->samples.begin(), h->samples.end(), [](const Link &a, const Link &b) { return a.name < b.name; });
  return ok;
}

static void walker_open(ppk_h5 *h) {
  h->fd = open(h->path.c_str(), O_RDONLY | O_CLOEXEC);
  if (h->fd < 0) throw std::string("cannot open ") + h->path + ": " + strerror(errno);
  struct stat st;
  if (fstat(h->fd, &st) != 0 || st.st_size < 96) throw Unsupported{"file too small"};
  void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, h->fd, 0);
  if (p == MAP_FAILED) throw Unsupported{"mmap failed"};
  h->map.p = (const uint8_t *)p;
  h->map.size = (size_t)st.st_size;
  const uint8_t *sb = h->map.p;
  static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  if (memcmp(sb, sig, 8) != 0) throw Unsupported{"no HDF5 signature at offset 0"};
  int ver = sb[8];
  if (ver > 1) throw Unsupported{"superblock version " + std::to_string(ver)};
  if (sb[13] != 8 || sb[14] != 8) throw Unsupported{"offsets / lengths are not 8 bytes"};
  size_t pos = ver == 0 ? 24 : 28;
  if (h->map.size < pos + 32 + 40) throw Unsupported{"file too small"};
  h->map.base = rd(sb + pos, 8);
  uint64_t eof = rd(sb + pos + 16, 8);
  if (h->map.base != 0) throw Unsupported{"non-zero base address"};
  if (eof > h->map.size) throw Unsupported{"file shorter than its end-of-file address (truncated?)"};
  const uint8_t *root = sb + pos + 32;
  uint64_t root_ohdr = rd(root + 8, 8), rb = UNDEF, rh = UNDEF;
  std::vector<Msg> scratch;
  group_of_object(h->map, root_ohdr, rb, rh, scratch);
  std::vector<Link> top;
  group_links(h->map, rb, rh, top, h->map.size / 40 + 16);
  const Link *sk = nullptr;
  for (const Link &l : top) {
    if (l.name == "sketches") sk = &l;
    if (l.name == "random") h->has_random = true;
  }
  if (!sk) throw std::string("no /sketches group in ") + h->path;
  h->sketches_ohdr = sk->ohdr;
  {
    std::vector<Msg> top_msgs;
    object_messages(h->map, sk->ohdr, top_msgs);
    for (const Msg &g : top_msgs) {
      Attr a;
      if (g.type == 0x000C && parse_attr(g, a) && a.count >= 1 && a.name == "codon_phased") h->codon_phased = attr_int(a, 0) != 0;
    }
  }
  uint64_t sb_ = UNDEF, sh_ = UNDEF;
  try {
    group_of_object(h->map, sk->ohdr, sb_, sh_, scratch);
    group_links(h->map, sb_, sh_, h->samples, h->map.size / 40 + 16);
  } catch (const Unsupported &u) {
    // a new-style /sketches: its listing through the library, the samples through this reader
    h->samples.clear();
    if (u.why.find("new-style") == std::string::npos || !hybrid_list_sketches(h)) throw;
    h->listed_by_library = true;
  }
  h->index.reserve(h->samples.size() * 2);
  for (size_t i = 0; i < h->samples.size(); i++) h->index.emplace(h->samples[i].name, i);
  walker_sample_params(h);
}

struct ReadJob {
  const ppk_h5 *h;
  const std::vector<const char *> *names;
  const std::vector<std::string> *knames;
  size_t words;
  uint64_t *out;
  int64_t *lengths, *missing;
  double *base_freq;
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::string error;          // first hard error (missing sample / k, wrong size)
  std::string unsupported;    // first structure the walker does not read
  std::atomic<bool> stop{false};
};

static void walker_read_range(ReadJob &job) {
  const ppk_h5 *h = job.h;
  const Map &m = h->map;
  std::vector<Msg> msgs, scratch;
  std::vector<Link> links;
  size_t nk = job.knames->size();
  const size_t batch = 16;
  for (;;) {
    size_t b0 = job.next.fetch_add(batch);
    if (b0 >= job.names->size() || job.stop.load(std::memory_order_relaxed)) return;
    size_t b1 = std::min(job.names->size(), b0 + batch);
    for (size_t i = b0; i < b1; i++) {
      try {
        auto it = h->index.find((*job.names)[i]);
        if (it == h->index.end()) throw std::string("sample ") + (*job.names)[i] + " not found in sketch database " + h->path;
        const Link &s = h->samples[it->second];
        uint64_t bt = UNDEF, hp = UNDEF;
        {
          // (the sample group's own header, never the addresses cached next to its name: see group_of_object)
          object_messages(m, s.ohdr, msgs);
          bool found = false;
          for (const Msg &g : msgs)
            if (g.type == 0x0011 && g.size >= 16 && !(g.flags & 2)) {
              bt = rd(g.data, 8);
              hp = rd(g.data + 8, 8);
              found = true;
            }
          if (!found) throw Unsupported{"sample group without a symbol table (new-style links)"};
          if (job.lengths) job.lengths[i] = 0;
          if (job.missing) job.missing[i] = 0;
          if (job.base_freq)
            for (int c = 0; c < 4; c++) job.base_freq[4 * i + c] = __builtin_nan("");
          for (const Msg &g : msgs) {
            if (g.type != 0x000C) continue;
            Attr a;
            if (!parse_attr(g, a) || a.count == 0) continue;
            if (job.lengths && a.name == "length") job.lengths[i] = attr_int(a, 0);
            if (job.missing && a.name == "missing_bases") job.missing[i] = attr_int(a, 0);
            if (job.base_freq && a.name == "base_freq" && a.count == 4)
              for (int c = 0; c < 4; c++) job.base_freq[4 * i + c] = attr_double(a, c);
          }
        }
        links.clear();
        group_links(m, bt, hp, links, 4096);
        for (size_t j = 0; j < nk; j++) {
          const Link *d = nullptr;
          for (const Link &l : links)
            if (l.name == (*job.knames)[j]) d = &l;
          if (!d)
            throw std::string("k-mer length ") + (*job.knames)[j] + " not found for sample " + s.name + " in " + h->path;
          uint64_t count = 0;
          Extent e = dataset_extent(m, d->ohdr, &count, scratch);
          if (count != job.words)
            throw std::string("sketch of ") + s.name + " at k=" + (*job.knames)[j] + " has " + std::to_string(count) +
                " words, expected sketchsize64*bbits = " + std::to_string(job.words);
          uint64_t *dst = job.out + (i * nk + j) * job.words;
          if (e.data)
            memcpy(dst, e.data, e.bytes);
          else
            memset(dst, 0, e.bytes);
        }
      } catch (const Unsupported &u) {
        std::lock_guard<std::mutex> g(job.mu);
        if (job.unsupported.empty()) job.unsupported = u.why;
        job.stop = true;
        return;
      } catch (const std::string &e) {
        std::lock_guard<std::mutex> g(job.mu);
        if (job.error.empty()) job.error = e;
        job.stop = true;
        return;
      }
    }
  }
}

// libhdf5 side ---------------------------------------------------------------------------------------------

static int lib_open(ppk_h5 *h, std::string &err) {
  std::lock_guard<std::mutex> g(g_h5_mutex);
  h->lib = hdf5_library(err);
  if (!h->lib) return PPK_ERR_STATE;
  Hdf5 *L = h->lib;
  h->file = L->H5Fopen(h->path.c_str(), 0 /* H5F_ACC_RDONLY */, 0);
  if (h->file < 0) {
    err = "libhdf5 cannot open " + h->path + (h->declined.empty() ? "" : " (direct reader: " + h->declined + ")");
    return PPK_ERR_ARG;
  }
  h->top = L->H5Gopen2(h->file, "sketches", 0);
  if (h->top < 0) {
    err = "no /sketches group in " + h->path;
    return PPK_ERR_ARG;
  }
  h->has_random = L->H5Lexists(h->file, "random", 0) > 0;
  if (L->H5Aexists(h->top, "codon_phased") > 0) {      // an enum (h5py bool) or an integer: read in its own type
    hid_t a = L->H5Aopen(h->top, "codon_phased", 0);
    hid_t t = a >= 0 ? L->H5Aget_type(a) : -1;
    uint64_t v = 0;
    if (t >= 0 && L->H5Tget_size(t) <= 8 && L->H5Aread(a, t, &v) >= 0) h->codon_phased = v != 0;
    if (t >= 0) L->H5Tclose(t);
    if (a >= 0) L->H5Aclose(a);
  }
  return PPK_OK;
}

static herr_t lib_collect(hid_t, const char *name, const void *, void *ud) {
  ((std::vector<std::string> *)ud)->push_back(name);
  return 0;
}

static void lib_names(ppk_h5 *h) {
  if (h->names5_read) return;
  std::lock_guard<std::mutex> g(g_h5_mutex);
  hsize_t idx = 0;
  h->lib->H5Literate(h->top, 0 /* H5_INDEX_NAME */, 0 /* H5_ITER_INC */, &idx, lib_collect, &h->names5);
  std::sort(h->names5.begin(), h->names5.end());
  h->names5_read = true;
}

static bool lib_attr(Hdf5 *L, hid_t obj, const char *name, hid_t type, void *out, size_t want, size_t *got) {
  if (L->H5Aexists(obj, name) <= 0) return false;
  hid_t a = L->H5Aopen(obj, name, 0);
  if (a < 0) return false;
  hid_t sp = L->H5Aget_space(a);
  long long n = sp >= 0 ? L->H5Sget_simple_extent_npoints(sp) : -1;
  if (sp >= 0) L->H5Sclose(sp);
  bool ok = false;
  if (n >= 0 && (size_t)n <= want) {
    ok = L->H5Aread(a, type, out) >= 0;
    if (got) *got = (size_t)n;
  }
  L->H5Aclose(a);
  return ok;
}

// The named sample's own parameters, into the caller's variables: nothing is inherited from another sample and
// nothing cached on the handle is touched (the handle caches the FIRST sample's parameters only, see ppk_h5_params).
static int lib_sample_params(ppk_h5 *h, const char *sample, size_t *s64, size_t *bbits, std::vector<int64_t> *kmers,
                             std::string &err) {
  std::lock_guard<std::mutex> g(g_h5_mutex);
  Hdf5 *L = h->lib;
  hid_t grp = L->H5Gopen2(h->top, sample, 0);
  if (grp < 0) {
    err = std::string("sample ") + sample + " not found in sketch database " + h->path;
    return PPK_ERR_ARG;
  }
  *s64 = *bbits = 0;
  kmers->clear();
  int64_t v = 0;
  if (lib_attr(L, grp, "sketchsize64", L->t_i64, &v, 1, nullptr)) *s64 = (size_t)v;
  if (lib_attr(L, grp, "bbits", L->t_i64, &v, 1, nullptr)) *bbits = (size_t)v;
  int64_t ks[256];
  size_t got = 0;
  if (lib_attr(L, grp, "kmers", L->t_i64, ks, 256, &got)) kmers->assign(ks, ks + got);
  L->H5Gclose(grp);
  if (*s64 == 0 || *bbits == 0) {
    err = std::string("sample ") + sample + " of " + h->path + " lacks sketchsize64 / bbits attributes";
    return PPK_ERR_ARG;
  }
  return PPK_OK;
}

static int lib_read(ppk_h5 *h, const std::vector<const char *> &names, const std::vector<std::string> &knames, size_t words,
                    uint64_t *out, int64_t *lengths, int64_t *missing, double *base_freq, std::string &err) {
  std::lock_guard<std::mutex> g(g_h5_mutex);
  Hdf5 *L = h->lib;
  size_t nk = knames.size();
  for (size_t i = 0; i < names.size(); i++) {
    hid_t grp = L->H5Gopen2(h->top, names[i], 0);
    if (grp < 0) {
      err = std::string("sample ") + names[i] + " not found in sketch database " + h->path;
      return PPK_ERR_ARG;
    }
    if (lengths) {
      lengths[i] = 0;
      lib_attr(L, grp, "length", L->t_i64, lengths + i, 1, nullptr);
    }
    if (missing) {
      missing[i] = 0;
      lib_attr(L, grp, "missing_bases", L->t_i64, missing + i, 1, nullptr);
    }
    if (base_freq) {
      double bf[4];
      size_t got = 0;
      bool ok = lib_attr(L, grp, "base_freq", L->t_f64, bf, 4, &got) && got == 4;
      for (int c = 0; c < 4; c++) base_freq[4 * i + c] = ok ? bf[c] : __builtin_nan("");
    }
    for (size_t j = 0; j < nk; j++) {
      hid_t d = L->H5Dopen2(grp, knames[j].c_str(), 0);
      if (d < 0) {
        L->H5Gclose(grp);
        err = "k-mer length " + knames[j] + " not found for sample " + names[i] + " in " + h->path;
        return PPK_ERR_ARG;
      }
      hid_t sp = L->H5Dget_space(d);
      long long cnt = sp >= 0 ? L->H5Sget_simple_extent_npoints(sp) : -1;
      if (sp >= 0) L->H5Sclose(sp);
      herr_t rc = -1;
      if (cnt == (long long)words) rc = L->H5Dread(d, L->t_u64, 0, 0, 0, out + (i * nk + j) * words);
      L->H5Dclose(d);
      if (cnt != (long long)words || rc < 0) {
        L->H5Gclose(grp);
        err = cnt != (long long)words
                  ? std::string("sketch of ") + names[i] + " at k=" + knames[j] + " has " + std::to_string(cnt) +
                        " words, expected sketchsize64*bbits = " + std::to_string(words)
                  : std::string("H5Dread failed for ") + names[i] + "/" + knames[j] + " in " + h->path;
        return PPK_ERR_ARG;
      }
    }
    L->H5Gclose(grp);
  }
  return PPK_OK;
}

static void close_handle(ppk_h5 *h) {
  if (!h) return;
  if (h->map.p) munmap((void *)h->map.p, h->map.size);
  if (h->fd >= 0) close(h->fd);
  if (h->lib) {
    std::lock_guard<std::mutex> g(g_h5_mutex);
    if (h->top >= 0) h->lib->H5Gclose(h->top);
    if (h->file >= 0) h->lib->H5Fclose(h->file);
  }
  delete h;
}

static void drop_walker(ppk_h5 *h) {
  if (h->map.p) munmap((void *)h->map.p, h->map.size);
  if (h->fd >= 0) close(h->fd);
  h->map = Map();
  h->fd = -1;
  h->samples.clear();
  h->index.clear();
}

}  // namespace

extern "C" {

int ppk_h5_set_library(const char *path) {
  std::lock_guard<std::mutex> g(g_h5_mutex);
  g_h5_hint = path ? path : "";
  return PPK_OK;
}

int ppk_h5_open(const char *path, int backend, ppk_h5 **out) {
  if (!path || !out || backend < 0 || backend > 2) return ppk_fail(PPK_ERR_ARG, "ppk_h5_open: bad argument");
  *out = nullptr;
  ppk_h5 *h = new ppk_h5;
  h->path = path;
  h->forced = backend != 0;
  if (backend != 2) {
    try {
      walker_open(h);
      h->backend = 1;
    } catch (const Unsupported &u) {
      h->declined = u.why;
      drop_walker(h);
    } catch (const std::string &e) {
      close_handle(h);
      return ppk_fail(PPK_ERR_ARG, "ppk_h5_open: " + e);
    }
    if (h->backend == 0 && backend == 1) {
      std::string why = h->declined;
      close_handle(h);
      return ppk_fail(PPK_ERR_STATE, "ppk_h5_open: the direct reader does not read " + std::string(path) + ": " + why);
    }
  }
  if (h->backend == 0) {
    std::string err;
    int rc = lib_open(h, err);
    if (rc != PPK_OK) {
      close_handle(h);
      return ppk_fail(rc, "ppk_h5_open: " + err);
    }
    h->backend = 2;
  }
  *out = h;
  return PPK_OK;
}

void ppk_h5_close(ppk_h5 *h) { close_handle(h); }

int ppk_h5_backend(const ppk_h5 *h) { return h ? h->backend : 0; }

int ppk_h5_has_random(const ppk_h5 *h) { return h && h->has_random ? 1 : 0; }

size_t ppk_h5_count(ppk_h5 *h) {
  if (!h) return 0;
  if (h->backend == 1) return h->samples.size();
  lib_names(h);
  return h->names5.size();
}

int ppk_h5_names(ppk_h5 *h, char *buf, size_t cap, size_t *need) {
  if (!h || !need) return ppk_fail(PPK_ERR_ARG, "ppk_h5_names: bad argument");
  if (h->backend == 2) lib_names(h);
  size_t n = h->backend == 1 ? h->samples.size() : h->names5.size();
  size_t bytes = 0;
  for (size_t i = 0; i < n; i++) bytes += (h->backend == 1 ? h->samples[i].name : h->names5[i]).size() + 1;
  *need = bytes;
  if (!buf || cap < bytes) return buf ? ppk_fail(PPK_ERR_CAPACITY, "ppk_h5_names: buffer too small") : PPK_OK;
  char *q = buf;
  for (size_t i = 0; i < n; i++) {
    const std::string &s = h->backend == 1 ? h->samples[i].name : h->names5[i];
    memcpy(q, s.c_str(), s.size() + 1);
    q += s.size() + 1;
  }
  return PPK_OK;
}

int ppk_h5_params(ppk_h5 *h, const char *sample, size_t *sketchsize64, size_t *bbits, int64_t *kmers, size_t kmers_cap,
                  size_t *n_kmers) {
  if (!h) return ppk_fail(PPK_ERR_ARG, "ppk_h5_params: bad argument");
  if (h->backend == 2 && sample) {
    // a named sample: its own parameters, straight to the caller
    size_t s64 = 0, bb = 0;
    std::vector<int64_t> ks;
    std::string err;
    int rc = lib_sample_params(h, sample, &s64, &bb, &ks, err);
    if (rc != PPK_OK) return ppk_fail(rc, "ppk_h5_params: " + err);
    if (sketchsize64) *sketchsize64 = s64;
    if (bbits) *bbits = bb;
    if (n_kmers) *n_kmers = ks.size();
    if (kmers)
      for (size_t i = 0; i < ks.size() && i < kmers_cap; i++) kmers[i] = ks[i];
    return PPK_OK;
  }
  if (h->backend == 2 && h->s64 == 0) {
    // no sample named: the first sample's, cached on the handle
    lib_names(h);
    if (h->names5.empty()) return ppk_fail(PPK_ERR_ARG, "ppk_h5_params: no samples in " + h->path);
    size_t s64 = 0, bb = 0;
    std::vector<int64_t> ks;
    std::string err;
    int rc = lib_sample_params(h, h->names5[0].c_str(), &s64, &bb, &ks, err);
    if (rc != PPK_OK) return ppk_fail(rc, "ppk_h5_params: " + err);
    h->s64 = s64;
    h->bbits = bb;
    h->kmers = ks;
  }
  if (h->backend == 1 && h->samples.empty()) return ppk_fail(PPK_ERR_ARG, "ppk_h5_params: no samples in " + h->path);
  if (h->backend == 1 && sample) {
    auto it = h->index.find(sample);
    if (it == h->index.end())
      return ppk_fail(PPK_ERR_ARG, std::string("ppk_h5_params: sample ") + sample + " not found in sketch database " + h->path);
    size_t s64 = 0, bb = 0;
    std::vector<int64_t> ks;
    try {
      sample_params_of(h, it->second, &s64, &bb, &ks);
    } catch (const Unsupported &u) {
      return ppk_fail(PPK_ERR_STATE, "ppk_h5_params: the direct reader does not read " + h->path + ": " + u.why);
    }
    if (sketchsize64) *sketchsize64 = s64;
    if (bbits) *bbits = bb;
    if (n_kmers) *n_kmers = ks.size();
    if (kmers)
      for (size_t i = 0; i < ks.size() && i < kmers_cap; i++) kmers[i] = ks[i];
    return PPK_OK;
  }
  if (sketchsize64) *sketchsize64 = h->s64;
  if (bbits) *bbits = h->bbits;
  if (n_kmers) *n_kmers = h->kmers.size();
  if (kmers)
    for (size_t i = 0; i < h->kmers.size() && i < kmers_cap; i++) kmers[i] = h->kmers[i];
  return PPK_OK;
}

int ppk_h5_read(ppk_h5 *h, const char *names, size_t n, const int32_t *kmers, size_t nk, size_t words, uint64_t *out,
                int64_t *lengths, int64_t *missing, double *base_freq, int threads) {
  if (!h || !names || !kmers || !out || nk == 0 || words == 0) return ppk_fail(PPK_ERR_ARG, "ppk_h5_read: bad argument");
  std::vector<const char *> list(n);
  const char *q = names;
  for (size_t i = 0; i < n; i++) {
    list[i] = q;
    q += strlen(q) + 1;
  }
  std::vector<std::string> knames(nk);
  for (size_t j = 0; j < nk; j++) knames[j] = std::to_string(kmers[j]);
  if (h->backend == 1) {
    ReadJob job;
    job.h = h;
    job.names = &list;
    job.knames = &knames;
    job.words = words;
    job.out = out;
    job.lengths = lengths;
    job.missing = missing;
    job.base_freq = base_freq;
    // the destination is fresh memory as a rule (np.empty): ask for huge pages before the threads touch it -- 45 faults
    // instead of 23 000 for a 90 MB array (the copy is bound by them, not by the decoding)
    {
      const uintptr_t a = ((uintptr_t)out + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)out + n * nk * words * 8) & ~(uintptr_t)4095;
      if (e > a + (4u << 20)) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
    }
    // most of the file is wanted: map its pages in one call instead of one fault per 4 KB from every thread
    // (MADV_POPULATE_READ, Linux 5.14; older kernels refuse the advice and the faults happen as the copy goes)
    if (n >= h->samples.size() / 4) madvise((void *)h->map.p, h->map.size, 22 /* MADV_POPULATE_READ */);
    unsigned hw = std::thread::hardware_concurrency();
    size_t t = threads > 0 ? (size_t)threads : std::min<size_t>(hw ? hw : 4, 16);
    t = std::max<size_t>(1, std::min(t, (n + 255) / 256));
    std::vector<std::thread> pool;
    for (size_t i = 1; i < t; i++) pool.emplace_back([&job] { walker_read_range(job); });
    walker_read_range(job);
    for (auto &th : pool) th.join();
    if (!job.error.empty()) return ppk_fail(PPK_ERR_ARG, "ppk_h5_read: " + job.error);
    if (job.unsupported.empty()) return PPK_OK;
    // an object the direct reader does not decode somewhere inside the file: the library reads all of it
    if (h->forced) return ppk_fail(PPK_ERR_STATE, "ppk_h5_read: the direct reader does not read " + h->path + ": " + job.unsupported);
    h->declined = job.unsupported;
    drop_walker(h);
    h->backend = 0;
    std::string err;
    int rc = lib_open(h, err);
    if (rc != PPK_OK) return ppk_fail(rc, "ppk_h5_read: " + err);
    h->backend = 2;
  }
  std::string err;
  int rc = lib_read(h, list, knames, words, out, lengths, missing, base_freq, err);
  if (rc != PPK_OK) return ppk_fail(rc, "ppk_h5_read: " + err);
  return PPK_OK;
}

const char *ppk_h5_declined(const ppk_h5 *h) { return h ? h->declined.c_str() : ""; }

int ppk_h5_codon_phased(const ppk_h5 *h) { return h ? h->codon_phased : -1; }

int ppk_h5_all_params(ppk_h5 *h, size_t count_cap, int64_t *sketchsize64, int64_t *bbits, int64_t *kmers, size_t kmers_cap,
                      size_t *n_kmers) {
  if (!h || !sketchsize64 || !bbits || !kmers || !n_kmers) return ppk_fail(PPK_ERR_ARG, "ppk_h5_all_params: bad argument");
  if (h->backend == 1) {
    if (h->samples.size() > count_cap)
      return ppk_fail(PPK_ERR_CAPACITY, "ppk_h5_all_params: the arrays hold fewer rows than the file has samples (ppk_h5_count)");
    try {
      std::vector<Msg> msgs;
      for (size_t i = 0; i < h->samples.size(); i++) {
        sketchsize64[i] = bbits[i] = 0;
        n_kmers[i] = 0;
        object_messages(h->map, h->samples[i].ohdr, msgs);
        for (const Msg &g : msgs) {
          Attr a;
          if (g.type != 0x000C || !parse_attr(g, a) || a.count == 0) continue;
          if (a.name == "sketchsize64") sketchsize64[i] = attr_int(a, 0);
          if (a.name == "bbits") bbits[i] = attr_int(a, 0);
          if (a.name == "kmers") {
            n_kmers[i] = a.count;
            for (uint64_t j = 0; j < a.count && j < kmers_cap; j++) kmers[i * kmers_cap + j] = attr_int(a, j);
          }
        }
      }
      return PPK_OK;
    } catch (const Unsupported &u) {
      if (h->forced) return ppk_fail(PPK_ERR_STATE, "ppk_h5_all_params: the direct reader does not read " + h->path + ": " + u.why);
      h->declined = u.why;
      drop_walker(h);
      h->backend = 0;
      std::string err;
      int rc = lib_open(h, err);
      if (rc != PPK_OK) return ppk_fail(rc, "ppk_h5_all_params: " + err);
      h->backend = 2;
    }
  }
  lib_names(h);
  // The caller sized its arrays from ppk_h5_count() -- possibly the direct reader's listing (hard links only), while
  // the library lists every link: after a hand-over the two counts can differ, and the rows must never outrun the
  // arrays.  The caller re-lists (ppk_h5_count / ppk_h5_names now answer from the library) and calls again.
  if (h->names5.size() > count_cap)
    return ppk_fail(PPK_ERR_CAPACITY, "ppk_h5_all_params: the sample list changed with the reader (re-list and call again)");
  std::lock_guard<std::mutex> g(g_h5_mutex);
  Hdf5 *L = h->lib;
  std::vector<int64_t> ks(4096);
  for (size_t i = 0; i < h->names5.size(); i++) {
    sketchsize64[i] = bbits[i] = 0;
    n_kmers[i] = 0;
    hid_t grp = L->H5Gopen2(h->top, h->names5[i].c_str(), 0);
    if (grp < 0) continue;
    lib_attr(L, grp, "sketchsize64", L->t_i64, sketchsize64 + i, 1, nullptr);
    lib_attr(L, grp, "bbits", L->t_i64, bbits + i, 1, nullptr);
    size_t got = 0;
    if (lib_attr(L, grp, "kmers", L->t_i64, ks.data(), ks.size(), &got)) {
      n_kmers[i] = got;
      for (size_t j = 0; j < got && j < kmers_cap; j++) kmers[i * kmers_cap + j] = ks[j];
    }
    L->H5Gclose(grp);
  }
  return PPK_OK;
}

}  // extern "C"
