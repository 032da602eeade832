// Host-side ingest for the packed-genome path: FASTA -> 2-bit + non-ACGT mask, BED -> site arrays, and the reference's
// segment / strand row order.  Counterparts in the reference (pure Python, >99 % of its wall time on real inputs):
//   SeqIO.to_dict(SeqIO.parse(ref_genome, 'fasta'))      MuRaL/data/preprocessing.py:836
//   bed_reader                                            MuRaL/data/preprocessing.py:39-106
//   the base maps of seq_digit_encoder / seq_ohe_encoder  MuRaL/data/preprocessing.py:655-666, :762-772
// Inputs may be gzip files (the reference's own prediction example feeds data/testing.bed.gz through pybedtools, which reads .gz
// transparently, MuRaL/scripts/run_predict.py:107): a file that starts with the gzip magic is inflated ONCE per process (zlib, every
// member of a multi-member / bgzip file) into an unlinked temporary file and every entry point below maps that copy.
// No device code here: the arrays produced feed mural_encode_* / mural_snv_forward_packed (include/mural_hip.h).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cctype>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// the bytes of one file, read-only: a mapping of the file itself or -- for a gzip file -- of its inflated copy
struct TextMap {
  const char* data = nullptr;
  size_t size = 0;
  int fd = -1;
  bool map_fd() {
    struct stat st;
    if (fstat(fd, &st) != 0) return false;
    size = (size_t)st.st_size;
    if (size == 0) return true;
    void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return false;
    data = static_cast<const char*>(p);
    return true;
  }
  ~TextMap() {
    if (data) munmap(const_cast<char*>(data), size);
    if (fd >= 0) ::close(fd);
  }
};

// inflate every gzip member of `src` into an unlinked temporary file ($TMPDIR, /tmp; anonymous memory if neither is writable)
int inflate_to_tmpfile(const TextMap& src, std::string* err) {
  int out = -1;
  const char* dirs[2] = {std::getenv("TMPDIR"), "/tmp"};
  for (const char* d : dirs) {
    if (!d || !*d) continue;
    std::string tmpl = std::string(d) + "/mural_gz_XXXXXX";
    out = mkstemp(&tmpl[0]);
    if (out >= 0) {
      unlink(tmpl.c_str());
      break;
    }
  }
  if (out < 0) out = memfd_create("mural_gz", 0);
  if (out < 0) {
    *err = "cannot create a temporary file for the inflated copy";
    return -1;
  }
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 16) != Z_OK) {
    *err = "zlib: inflateInit2 failed";
    ::close(out);
    return -1;
  }
  std::vector<unsigned char> buf(4u << 20);
  const unsigned char* in = reinterpret_cast<const unsigned char*>(src.data);
  size_t left = src.size;
  bool ok = true, member_open = false;
  while (ok && left > 0) {
    if (!member_open) {      // between members: zero padding is legal behind the last one
      size_t z = 0;
      while (z < left && in[z] == 0) ++z;
      if (z == left) break;
      in += z;
      left -= z;
      member_open = true;
    }
    const size_t feed = std::min<size_t>(left, 1u << 30);
    zs.next_in = const_cast<unsigned char*>(in);
    zs.avail_in = (uInt)feed;
    int rc = Z_OK;
    do {
      zs.next_out = buf.data();
      zs.avail_out = (uInt)buf.size();
      rc = inflate(&zs, Z_NO_FLUSH);
      if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
        *err = std::string("zlib: ") + (zs.msg ? zs.msg : "corrupt gzip stream");
        ok = false;
        break;
      }
      const size_t have = buf.size() - zs.avail_out;
      size_t off = 0;
      while (off < have) {
        const ssize_t w = ::write(out, buf.data() + off, have - off);
        if (w <= 0) {
          *err = "cannot write the inflated copy (temporary directory full?)";
          ok = false;
          break;
        }
        off += (size_t)w;
      }
    } while (ok && rc != Z_STREAM_END && (zs.avail_out == 0 || zs.avail_in > 0));
    const size_t used = feed - zs.avail_in;
    in += used;
    left -= used;
    if (ok && rc == Z_STREAM_END) {
      member_open = false;
      inflateReset(&zs);
    } else if (ok && used == 0 && rc == Z_BUF_ERROR) {
      *err = "zlib: truncated gzip stream";
      ok = false;
    }
  }
  if (ok && member_open) {
    *err = "zlib: truncated gzip stream";
    ok = false;
  }
  inflateEnd(&zs);
  if (!ok) {
    ::close(out);
    return -1;
  }
  return out;
}

// inflated copies are kept per process (a FASTA record is packed by one call per record: re-inflating a genome per call would
// cost seconds each); an entry is dropped when the file changes or when four newer ones exist
struct GzEntry {
  std::string path;
  dev_t dev;
  ino_t ino;
  off_t size;
  struct timespec mtime;
  std::shared_ptr<TextMap> text;
};
std::mutex g_gz_mutex;
std::vector<GzEntry> g_gz_cache;

// nullptr + *err on failure
std::shared_ptr<TextMap> open_text(const char* path, std::string* err) {
  auto raw = std::make_shared<TextMap>();
  raw->fd = ::open(path, O_RDONLY);
  struct stat st;
  if (raw->fd < 0 || fstat(raw->fd, &st) != 0 || !raw->map_fd()) {
    *err = "cannot open";
    return nullptr;
  }
  const bool gz = raw->size >= 2 && (unsigned char)raw->data[0] == 0x1f && (unsigned char)raw->data[1] == 0x8b;
  if (!gz) return raw;      // (no MADV_SEQUENTIAL: every reader makes two passes, and pages dropped behind the first are faulted in again)
  std::lock_guard<std::mutex> lock(g_gz_mutex);
  for (size_t i = 0; i < g_gz_cache.size(); ++i) {
    GzEntry& e = g_gz_cache[i];
    if (e.dev == st.st_dev && e.ino == st.st_ino) {
      if (e.size == st.st_size && e.mtime.tv_sec == st.st_mtim.tv_sec && e.mtime.tv_nsec == st.st_mtim.tv_nsec) {
        GzEntry hit = e;
        g_gz_cache.erase(g_gz_cache.begin() + (long)i);
        g_gz_cache.push_back(hit);
        return hit.text;
      }
      g_gz_cache.erase(g_gz_cache.begin() + (long)i);
      break;
    }
  }
  auto text = std::make_shared<TextMap>();
  text->fd = inflate_to_tmpfile(*raw, err);
  if (text->fd < 0) return nullptr;
  if (!text->map_fd()) {
    *err = "cannot map the inflated copy";
    return nullptr;
  }
  if (g_gz_cache.size() >= 4) g_gz_cache.erase(g_gz_cache.begin());
  g_gz_cache.push_back(GzEntry{path, st.st_dev, st.st_ino, st.st_size, st.st_mtim, text});
  return text;
}

// the entry points' view of a file (kept alive by the shared mapping)
struct MappedFile {
  const char* data = nullptr;
  size_t size = 0;
  std::shared_ptr<TextMap> hold;
  std::string why;
  bool open(const char* path) {
    hold = open_text(path, &why);
    if (!hold) return false;
    data = hold->data;
    size = hold->size;
    return true;
  }
};

// MURAL_SYM_* of a character (either case): 0..3 = A C G T, 4 = N, 5..14 = R Y M S W K B D H V; 254 = white space (isspace in the C
// locale), 255 = not a nucleotide code
struct BaseTable {
  uint8_t t[256];
  BaseTable() {
    std::memset(t, 255, sizeof(t));
    for (const char* w = " \t\n\v\f\r"; *w; ++w) t[(unsigned char)*w] = 254;
    const char* alphabet = "ACGTNRYMSWKBDHV";
    for (int i = 0; alphabet[i]; ++i) {
      t[(unsigned char)alphabet[i]] = (uint8_t)i;
      t[(unsigned char)std::tolower(alphabet[i])] = (uint8_t)i;
    }
  }
};
const BaseTable kBase;

int host_threads() {
  if (const char* e = std::getenv("MURAL_HOST_THREADS")) {
    const int v = std::atoi(e);
    if (v >= 1) return std::min(v, 256);
  }
  return (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
}

template <typename Fn>
void run_parallel(int T, Fn fn) {
  if (T == 1) {
    fn(0);
    return;
  }
  std::vector<std::thread> th;
  for (int k = 0; k < T; ++k) th.emplace_back(fn, k);
  for (auto& x : th) x.join();
}

inline const char* line_end(const char* p, const char* end) {
  const void* q = std::memchr(p, '\n', (size_t)(end - p));
  return q ? static_cast<const char*>(q) : end;
}

}  // namespace

using namespace mural;

// Scan a FASTA file: record names (text after '>' up to the first whitespace, like Bio.SeqIO ids), sequence lengths
// (whitespace stripped) and the byte offset of each record's first sequence line.  names: n_cap x name_cap chars, NUL-terminated.
extern "C" int mural_fasta_scan(const char* path, int64_t n_cap, int32_t name_cap, char* names, int64_t* lengths,
                                int64_t* offsets, int64_t* n_records) {
  MURAL_REQUIRE(path && n_records, "NULL argument");
  MappedFile f;
  if (!f.open(path)) {
    set_error("cannot open FASTA file %s%s%s", path, f.why.empty() || f.why == "cannot open" ? "" : ": ", f.why == "cannot open" ? "" : f.why.c_str());
    return MURAL_E_INVALID;
  }
  // A genome is gigabytes of text and every rank scans it: the file is cut into byte chunks at line starts, every host thread lists
  // the header lines of its chunk and counts the bases in front of its first header and behind each header; the pieces add up.
  struct Hdr { const char* line; const char* line_end; int64_t bases; };
  struct Part { int64_t lead = 0; std::vector<Hdr> hdr; };
  const char* end = f.data + f.size;
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), f.size / (8u << 20) + 1));
  std::vector<const char*> cut((size_t)T + 1);
  cut[0] = f.data;
  for (int k = 1; k < T; ++k) {
    const char* at = f.data + f.size * (size_t)k / (size_t)T;
    if (at > f.data && at[-1] != '\n') {
      const char* e = line_end(at, end);
      at = e < end ? e + 1 : end;
    }
    cut[(size_t)k] = std::max(at, cut[(size_t)k - 1]);
  }
  cut[(size_t)T] = end;
  std::vector<Part> part((size_t)T);
  run_parallel(T, [&](int k) {
    Part& P = part[(size_t)k];
    int64_t* count = &P.lead;
    const char* p = cut[(size_t)k];
    const char* hi = cut[(size_t)k + 1];
    while (p < hi) {
      const char* e = line_end(p, hi);
      if (*p == '>') {
        P.hdr.push_back(Hdr{p, e, 0});
        count = &P.hdr.back().bases;
      } else {
        int64_t n = 0;
        for (const char* q = p; q < e; ++q) n += kBase.t[(unsigned char)*q] != 254;
        *count += n;
      }
      p = e < hi ? e + 1 : hi;
    }
  });
  int64_t n = 0;
  bool in_record = false;
  for (auto& P : part) {
    if (P.lead) {
      if (!in_record) {
        set_error("%s: sequence data before the first '>' header", path);
        return MURAL_E_INVALID;
      }
      if (n <= n_cap && lengths) lengths[n - 1] += P.lead;
    }
    for (auto& h : P.hdr) {
      ++n;
      in_record = true;
      if (n <= n_cap) {
        if (names) {
          const char* q = h.line + 1;
          int k = 0;
          while (q < h.line_end && !std::isspace((unsigned char)*q) && k + 1 < name_cap) names[(n - 1) * (int64_t)name_cap + k++] = *q++;
          names[(n - 1) * (int64_t)name_cap + k] = '\0';
        }
        if (offsets) offsets[n - 1] = (int64_t)((h.line_end < end ? h.line_end + 1 : end) - f.data);
        if (lengths) lengths[n - 1] = h.bases;
      }
    }
  }
  *n_records = n;
  return MURAL_OK;
}

// Pack one record (starting at byte `offset`, `length` bases) into the device format of include/mural_hip.h:
// packed2: ceil(length/16) words (A0 C1 G2 T3, 2 bits per base), nmask: ceil(length/32) words (bit set = not ACGT).
// Positions and symbols of IUPAC codes other than N are reported in amb_pos / amb_sym (up to amb_cap; n_amb counts all of
// them): the side table of MuralGenome (fractional one-hot columns, preprocessing.py:762-772).  Any other character is an
// error, like the reference's dict lookups (KeyError).
extern "C" int mural_fasta_pack(const char* path, int64_t offset, int64_t length, uint32_t* packed2, uint32_t* nmask,
                                int64_t* amb_pos, uint8_t* amb_sym, int64_t amb_cap, int64_t* n_amb) {
  MURAL_REQUIRE(path && packed2 && nmask, "NULL argument");
  MappedFile f;
  if (!f.open(path)) {
    set_error("cannot open FASTA file %s%s%s", path, f.why.empty() || f.why == "cannot open" ? "" : ": ", f.why == "cannot open" ? "" : f.why.c_str());
    return MURAL_E_INVALID;
  }
  MURAL_REQUIRE(offset >= 0 && (size_t)offset <= f.size, "record offset outside the file");
  std::memset(packed2, 0, (size_t)((length + 15) / 16) * 4);
  std::memset(nmask, 0, (size_t)((length + 31) / 32) * 4);
  const char* lo = f.data + offset;
  const char* file_end = f.data + f.size;
  // A human chromosome is 250 MB of text and N ranks each pack it: the record's bytes are cut into chunks, pass 1 counts the bases of
  // every chunk (its first base index) and looks for the next record's '>', pass 2 packs the chunks side by side -- a chunk owns every
  // 32-base word that lies wholly inside its base range and ORs atomically into the two it may share with its neighbours.  Where the
  // record ends is not known up front: the chunks cover the bytes `length` bases take at the width of the record's first line
  // (+ slack); a record with ragged lines that runs past them is finished by a serial scan.
  const char* cap_end = file_end;
  if (lo < file_end) {
    const char* nl = static_cast<const char*>(std::memchr(lo, '\n', std::min<size_t>((size_t)(file_end - lo), 1u << 20)));
    if (nl) {
      int64_t on_line = 0;
      for (const char* q = lo; q < nl; ++q) on_line += kBase.t[(unsigned char)*q] != 254;
      if (on_line > 0) {
        const size_t est = (size_t)(length / on_line + 2) * (size_t)(nl - lo + 1) + 4096;
        if (est < (size_t)(file_end - lo)) cap_end = lo + est;
      }
    }
  }
  const size_t span = (size_t)(cap_end - lo);
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), span / (8u << 20) + 1));
  std::vector<const char*> cut((size_t)T + 2);
  for (int k = 0; k <= T; ++k) cut[(size_t)k] = lo + span * (size_t)k / (size_t)T;
  std::vector<int64_t> first((size_t)T + 2, 0);
  std::vector<const char*> stop((size_t)T + 1, nullptr);      // the '>' a chunk ran into
  run_parallel(T, [&](int k) {
    int64_t n = 0;
    const char* p = cut[(size_t)k];
    for (; p < cut[(size_t)k + 1] && *p != '>'; ++p) n += kBase.t[(unsigned char)*p] != 254;
    first[(size_t)k + 1] = n;
    if (p < cut[(size_t)k + 1]) stop[(size_t)k] = p;
  });
  int chunks = T;
  for (int k = 0; k < T; ++k)
    if (stop[(size_t)k]) {                 // the record ends inside chunk k
      cut[(size_t)k + 1] = stop[(size_t)k];
      chunks = k + 1;
      break;
    }
  for (int k = 0; k < chunks; ++k) first[(size_t)k + 1] += first[(size_t)k];
  if (chunks == T && !stop[(size_t)T - 1] && cap_end < file_end && first[(size_t)T] < length) {      // ragged lines: one more, serial chunk
    const void* gt = std::memchr(cap_end, '>', (size_t)(file_end - cap_end));
    cut[(size_t)T + 1] = gt ? static_cast<const char*>(gt) : file_end;
    int64_t n = 0;
    for (const char* p = cap_end; p < cut[(size_t)T + 1]; ++p) n += kBase.t[(unsigned char)*p] != 254;
    first[(size_t)T + 1] = first[(size_t)T] + n;
    chunks = T + 1;
  }
  if (first[(size_t)chunks] != length) {
    set_error("%s: record holds %lld bases, the scan announced %lld", path, (long long)first[(size_t)chunks], (long long)length);
    return MURAL_E_INVALID;
  }
  struct Amb { int64_t pos; uint8_t sym; };
  std::vector<std::vector<Amb>> ambs((size_t)chunks);
  std::vector<std::string> errors((size_t)chunks);
  run_parallel(chunks, [&](int k) {
    int64_t i = first[(size_t)k];
    const int64_t i_end = first[(size_t)k + 1];
    if (i == i_end) return;
    const int64_t w_first = i >> 5, w_last = (i_end - 1) >> 5;       // 32-base words this chunk may share
    uint32_t pw[2] = {0u, 0u}, mw = 0u;                                // the two packed words and the mask word of 32-base word i >> 5
    auto flush = [&](int64_t w) {
      if (w == w_first || w == w_last) {
        if (pw[0]) __atomic_fetch_or(&packed2[2 * w], pw[0], __ATOMIC_RELAXED);
        if (pw[1] && 2 * w + 1 < (length + 15) / 16) __atomic_fetch_or(&packed2[2 * w + 1], pw[1], __ATOMIC_RELAXED);
        if (mw) __atomic_fetch_or(&nmask[w], mw, __ATOMIC_RELAXED);
      } else {
        packed2[2 * w] = pw[0];
        packed2[2 * w + 1] = pw[1];
        nmask[w] = mw;
      }
      pw[0] = pw[1] = mw = 0u;
    };
    for (const char* p = cut[(size_t)k]; p < cut[(size_t)k + 1]; ++p) {
      const unsigned char ch = (unsigned char)*p;
      const uint8_t code = kBase.t[ch];
      if (code == 254) continue;
      if (code == 255) {
        char msg[512];
        std::snprintf(msg, sizeof(msg), "%s: character '%c' at base %lld is not a nucleotide code", path, ch, (long long)i);
        errors[(size_t)k] = msg;
        return;
      }
      const int b = (int)(i & 31);
      if (code < 4) {
        pw[b >> 4] |= (uint32_t)code << (2 * (b & 15));
      } else {
        mw |= 1u << b;
        if (code > 4) ambs[(size_t)k].push_back(Amb{i, code});
      }
      ++i;
      if ((i & 31) == 0) flush((i - 1) >> 5);
    }
    if (i & 31) flush(i >> 5);
  });
  for (auto& e : errors)
    if (!e.empty()) {
      set_error("%s", e.c_str());
      return MURAL_E_INVALID;
    }
  int64_t amb = 0;
  for (auto& v : ambs)
    for (auto& a : v) {
      if (amb < amb_cap) {
        if (amb_pos) amb_pos[amb] = a.pos;
        if (amb_sym) amb_sym[amb] = a.sym;
      }
      ++amb;
    }
  if (n_amb) *n_amb = amb;
  return MURAL_OK;
}

// Read a BED file with the reference's six columns (chrom, start, end, name, score = class label, strand).  Chromosome
// names are interned in order of first appearance: chrom_id indexes `chrom_names` (n_chrom_cap x name_cap).  Call with
// cap = 0 to count rows and chromosomes first.  Header / track / comment lines are skipped like pybedtools does.
// The file is cut into byte ranges at line boundaries and parsed by up to 16 host threads (MURAL_HOST_THREADS overrides): a
// whole-genome site list is gigabytes of text, and one core parses ~150 MB/s.
namespace {

struct BedChunk {
  const char* lo = nullptr;
  const char* hi = nullptr;
  int64_t first_line = 0, rows = 0, row0 = 0, lines = 0;
  std::vector<std::string> chroms;     // local interning, order of first appearance inside the chunk
  std::vector<int32_t> remap;          // local id -> global id
  std::string error;
};

inline bool bed_skip_line(const char* p, size_t len) {
  return len == 0 || *p == '#' || (len >= 5 && !std::strncmp(p, "track", 5)) || (len >= 7 && !std::strncmp(p, "browser", 7));
}

// decimal integer spanning exactly [f, f + len); the common all-digit case without strtoll
inline bool parse_i64_field(const char* f, size_t len, long long* out) {
  if (len == 0) return false;
  if (len <= 18) {
    long long v = 0;
    size_t i = 0;
    for (; i < len; ++i) {
      const unsigned d = (unsigned)(f[i] - '0');
      if (d > 9) break;
      v = v * 10 + d;
    }
    if (i == len) {
      *out = v;
      return true;
    }
  }
  char* stop = nullptr;
  *out = std::strtoll(f, &stop, 10);
  return stop == f + len;
}

inline bool parse_score_field(const char* f, size_t len, float* out) {
  if (len == 1 && (unsigned)(f[0] - '0') <= 9) {      // class labels are single digits
    *out = (float)(f[0] - '0');
    return true;
  }
  char* stop = nullptr;
  *out = std::strtof(f, &stop);
  return stop == f + len && len > 0;
}

// pass 1 (fill = false): count rows and lines; pass 2: parse into the arrays at row0
void bed_parse_chunk(BedChunk& c, const char* path, bool fill, int32_t* chrom_id, int64_t* start, int64_t* end_, float* score,
                     uint8_t* strand) {
  const char* p = c.lo;
  int64_t n = 0, line_no = c.first_line;
  int last = -1;
  while (p < c.hi) {
    const char* e = line_end(p, c.hi);
    ++line_no;
    const char* le = e;
    while (le > p && (le[-1] == '\r' || le[-1] == ' ' || le[-1] == '\t')) --le;
    const size_t len = (size_t)(le - p);
    if (!bed_skip_line(p, len)) {
      if (fill) {
        const char* fld[6];
        size_t flen[6];
        int nf = 0;
        const char* q = p;
        while (q <= le && nf < 6) {
          const char* t = static_cast<const char*>(std::memchr(q, '\t', (size_t)(le - q)));
          if (!t) t = le;
          fld[nf] = q;
          flen[nf] = (size_t)(t - q);
          ++nf;
          q = t + 1;
        }
        char msg[512];
        if (nf < 6) {
          std::snprintf(msg, sizeof(msg), "%s:%lld: expected 6 tab-separated BED columns (chrom start end name score strand), got %d", path,
                        (long long)line_no, nf);
          c.error = msg;
          return;
        }
        long long s = 0, t2 = 0;
        float sc = 0.f;
        const bool ok = parse_i64_field(fld[1], flen[1], &s) && parse_i64_field(fld[2], flen[2], &t2) &&
                        parse_score_field(fld[4], flen[4], &sc) && flen[5] == 1 && (fld[5][0] == '+' || fld[5][0] == '-');
        if (!ok || s < 0 || t2 < s) {
          std::snprintf(msg, sizeof(msg), "%s:%lld: malformed BED row", path, (long long)line_no);
          c.error = msg;
          return;
        }
        int cid = -1;
        if (last >= 0 && c.chroms[(size_t)last].size() == flen[0] && !std::memcmp(c.chroms[(size_t)last].data(), fld[0], flen[0])) cid = last;
        for (int k = 0; cid < 0 && k < (int)c.chroms.size(); ++k)
          if (c.chroms[(size_t)k].size() == flen[0] && !std::memcmp(c.chroms[(size_t)k].data(), fld[0], flen[0])) cid = k;
        if (cid < 0) {
          cid = (int)c.chroms.size();
          c.chroms.emplace_back(fld[0], flen[0]);
        }
        last = cid;
        const int64_t r = c.row0 + n;
        chrom_id[r] = cid;
        start[r] = s;
        end_[r] = t2;
        score[r] = sc;
        strand[r] = fld[5][0] == '-' ? 1 : 0;
      }
      ++n;
    }
    p = e < c.hi ? e + 1 : c.hi;
  }
  c.rows = n;
  c.lines = line_no - c.first_line;
}

}  // namespace

extern "C" int mural_bed_read(const char* path, int64_t cap, int32_t* chrom_id, int64_t* start, int64_t* end_, float* score,
                              uint8_t* strand, int32_t n_chrom_cap, int32_t name_cap, char* chrom_names, int64_t* n_rows,
                              int32_t* n_chroms) {
  MURAL_REQUIRE(path && n_rows && n_chroms, "NULL argument");
  MappedFile f;
  if (!f.open(path)) {
    set_error("cannot open BED file %s%s%s", path, f.why.empty() || f.why == "cannot open" ? "" : ": ", f.why == "cannot open" ? "" : f.why.c_str());
    return MURAL_E_INVALID;
  }
  const char* end = f.data + f.size;
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), f.size / (4u << 20) + 1));
  std::vector<BedChunk> ch((size_t)T);
  const char* p = f.data;
  for (int k = 0; k < T; ++k) {
    ch[(size_t)k].lo = p;
    const char* cut = k + 1 == T ? end : f.data + f.size * (size_t)(k + 1) / (size_t)T;
    if (cut < p) cut = p;
    if (cut < end) {
      const char* e = line_end(cut, end);
      cut = e < end ? e + 1 : end;
    }
    ch[(size_t)k].hi = cut;
    p = cut;
  }
  // pass 1: rows per chunk (newline scan), so that every chunk knows where its rows go
  run_parallel(T, [&](int k) { bed_parse_chunk(ch[(size_t)k], path, false, nullptr, nullptr, nullptr, nullptr, nullptr); });
  int64_t n = 0, line0 = 0;
  for (auto& c : ch) {
    c.row0 = n;
    c.first_line = line0;
    n += c.rows;
    line0 += c.lines;
  }
  *n_rows = n;
  const bool fill = cap >= n && n > 0 && chrom_id && start && end_ && score && strand;
  if (!fill) {
    // counting call: only the chromosome names are still needed -- first fields only, no arrays
    run_parallel(T, [&](int k) {
      BedChunk& c = ch[(size_t)k];
      const char* q = c.lo;
      int last = -1;
      while (q < c.hi) {
        const char* e = line_end(q, c.hi);
        const char* le = e;
        while (le > q && (le[-1] == '\r' || le[-1] == ' ' || le[-1] == '\t')) --le;
        const size_t len = (size_t)(le - q);
        if (!bed_skip_line(q, len)) {
          const char* t = static_cast<const char*>(std::memchr(q, '\t', len));
          const size_t fl = t ? (size_t)(t - q) : len;
          int cid = -1;
          if (last >= 0 && c.chroms[(size_t)last].size() == fl && !std::memcmp(c.chroms[(size_t)last].data(), q, fl)) cid = last;
          for (int j = 0; cid < 0 && j < (int)c.chroms.size(); ++j)
            if (c.chroms[(size_t)j].size() == fl && !std::memcmp(c.chroms[(size_t)j].data(), q, fl)) cid = j;
          if (cid < 0) {
            cid = (int)c.chroms.size();
            c.chroms.emplace_back(q, fl);
          }
          last = cid;
        }
        q = e < c.hi ? e + 1 : c.hi;
      }
    });
    std::vector<std::string> chroms;
    for (auto& c : ch)
      for (auto& name : c.chroms)
        if (std::find(chroms.begin(), chroms.end(), name) == chroms.end()) chroms.push_back(name);
    *n_chroms = (int32_t)chroms.size();
    return MURAL_OK;
  }
  run_parallel(T, [&](int k) { bed_parse_chunk(ch[(size_t)k], path, true, chrom_id, start, end_, score, strand); });
  for (auto& c : ch)
    if (!c.error.empty()) {
      set_error("%s", c.error.c_str());
      return MURAL_E_INVALID;
    }
  // global interning in file order (chunks are in file order, local lists in order of first appearance), then the id remap
  std::vector<std::string> chroms;
  for (auto& c : ch) {
    c.remap.resize(c.chroms.size());
    for (size_t k = 0; k < c.chroms.size(); ++k) {
      auto it = std::find(chroms.begin(), chroms.end(), c.chroms[k]);
      if (it == chroms.end()) {
        c.remap[k] = (int32_t)chroms.size();
        chroms.push_back(c.chroms[k]);
      } else {
        c.remap[k] = (int32_t)(it - chroms.begin());
      }
    }
  }
  run_parallel(T, [&](int k) {
    BedChunk& c = ch[(size_t)k];
    bool identity = true;
    for (size_t j = 0; j < c.remap.size(); ++j) identity = identity && c.remap[j] == (int32_t)j;
    if (identity) return;
    for (int64_t r = c.row0; r < c.row0 + c.rows; ++r) chrom_id[r] = c.remap[(size_t)chrom_id[r]];
  });
  *n_chroms = (int32_t)chroms.size();
  if (chrom_names)
    for (int k = 0; k < (int)chroms.size() && k < n_chrom_cap; ++k) {
      std::strncpy(chrom_names + (int64_t)k * name_cap, chroms[(size_t)k].c_str(), (size_t)name_cap - 1);
      chrom_names[(int64_t)k * name_cap + name_cap - 1] = '\0';
    }
  return MURAL_OK;
}

// ---- rank-local, streaming BED ingest (whole-genome prediction on N ranks: mural_amd.predict.predict_bed_sharded) ------------------
// The reference parses one whole BED per process and advises to split big inputs into ~1 M-site files by hand
// (MuRaL/commands/predict.py:134-137, MuRaL/scripts/run_predict.py:107).  Here the file is INDEXED once -- every rank scans 1 / N of
// its bytes (mural_bed_index_scan) and the ranks exchange the pieces they found -- and afterwards a rank parses only the rows of its
// own block of a chromosome (mural_bed_parse_range): host memory is bounded by one rank's share of one chromosome.
//
// A piece = consecutive rows of ONE chromosome: name, byte range [lo, hi) (from the first byte of its first row to the first byte
// behind its last row's line), row count, start of its first row.  A piece never holds more than piece_rows rows, so that row r of
// a chromosome is found by a newline scan of at most one piece.  The scan covers the lines that START in [byte_lo, byte_hi)
// (byte_lo is moved to the next line start unless it is one); comment / track / browser / blank lines belong to no piece.
// Call with byte_lo = byte_hi = 0 to learn file_bytes (the inflated size of a .gz file).  n_pieces may exceed cap: call again.
namespace {

struct BedPiece {
  std::string name;
  int64_t lo, hi, rows, first_start;
  // table order inside the piece: every row's (start, strand) is >= its predecessor's ('+' < '-') -- then the reference's output order
  // (bed_reader's segments, '+' rows before '-' rows, stably re-sorted by start: run_predict.py:227) IS the file order
  int64_t in_order, last_start, first_strand, last_strand;
};

void bed_index_chunk(const char* base, const char* lo, const char* hi, const char* file_end, int64_t piece_rows,
                     std::vector<BedPiece>& out, std::string& error, const char* path) {
  const char* p = lo;
  BedPiece cur{std::string(), 0, 0, 0, 0, 1, 0, 0, 0};
  auto flush = [&]() {
    if (cur.rows) out.push_back(cur);
    cur.rows = 0;
  };
  while (p < hi) {                                             // (a line that starts in front of hi is scanned to its end)
    const char* e = line_end(p, file_end);
    const char* le = e;
    while (le > p && (le[-1] == '\r' || le[-1] == ' ' || le[-1] == '\t')) --le;
    const size_t len = (size_t)(le - p);
    const char* next = e < file_end ? e + 1 : file_end;
    if (!bed_skip_line(p, len)) {
      const char* t = static_cast<const char*>(std::memchr(p, '\t', len));
      const char* t2 = t ? static_cast<const char*>(std::memchr(t + 1, '\t', (size_t)(le - t - 1))) : nullptr;
      long long st = 0;
      if (!t || !t2 || !parse_i64_field(t + 1, (size_t)(t2 - t - 1), &st) || st < 0) {
        char msg[512];
        std::snprintf(msg, sizeof(msg), "%s: malformed BED row at byte offset %lld", path, (long long)(p - base));
        error = msg;
        return;
      }
      // strand = the sixth field ('+' / '-'; a row without one is reported by the parser later: here it only ends the "in order" claim)
      int sd = -1;
      {
        const char* q = t2;
        for (int k = 0; k < 3 && q; ++k) q = static_cast<const char*>(std::memchr(q + 1, '\t', (size_t)(le - q - 1)));
        if (q && q + 1 < le && (q + 2 == le || q[2] == '\t') && (q[1] == '+' || q[1] == '-')) sd = q[1] == '-' ? 1 : 0;
      }
      const size_t fl = (size_t)(t - p);
      if (cur.rows && (cur.rows >= piece_rows || cur.name.size() != fl || std::memcmp(cur.name.data(), p, fl) != 0)) flush();
      if (!cur.rows) {
        cur.name.assign(p, fl);
        cur.lo = (int64_t)(p - base);
        cur.first_start = st;
        cur.first_strand = sd;
        cur.in_order = sd >= 0;
      } else if (sd < 0 || st < cur.last_start || (st == cur.last_start && sd < cur.last_strand)) {
        cur.in_order = 0;
      }
      cur.last_start = st;
      cur.last_strand = sd;
      ++cur.rows;
      cur.hi = (int64_t)(next - base);
    }
    p = next;
  }
  flush();
}

// first line start at or behind `at`
const char* align_line(const char* base, const char* at, const char* end) {
  if (at <= base) return base;
  if (at >= end) return end;
  if (at[-1] == '\n') return at;
  const char* e = line_end(at, end);
  return e < end ? e + 1 : end;
}

}  // namespace

extern "C" int mural_bed_index_scan(const char* path, int64_t byte_lo, int64_t byte_hi, int64_t piece_rows, int32_t name_cap, int64_t cap,
                                    char* names, int64_t* p_lo, int64_t* p_hi, int64_t* p_rows, int64_t* p_first_start,
                                    int64_t* p_order, int64_t* n_pieces, int64_t* file_bytes) {
  MURAL_REQUIRE(path && n_pieces && file_bytes, "NULL argument");
  MURAL_REQUIRE(piece_rows >= 1 && byte_lo >= 0 && byte_hi >= byte_lo, "bad scan range");
  MappedFile f;
  if (!f.open(path)) {
    set_error("cannot open BED file %s%s%s", path, f.why == "cannot open" ? "" : ": ", f.why == "cannot open" ? "" : f.why.c_str());
    return MURAL_E_INVALID;
  }
  *file_bytes = (int64_t)f.size;
  *n_pieces = 0;
  const char* end = f.data + f.size;
  const char* lo = align_line(f.data, f.data + std::min<size_t>((size_t)byte_lo, f.size), end);
  const char* hi = align_line(f.data, f.data + std::min<size_t>((size_t)byte_hi, f.size), end);
  if (lo >= hi) return MURAL_OK;
  const size_t span = (size_t)(hi - lo);
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), span / (4u << 20) + 1));
  std::vector<std::vector<BedPiece>> found((size_t)T);
  std::vector<std::string> errors((size_t)T);
  std::vector<const char*> cut((size_t)T + 1);
  cut[0] = lo;
  for (int k = 1; k < T; ++k) cut[(size_t)k] = std::max(cut[(size_t)k - 1], align_line(f.data, lo + span * (size_t)k / (size_t)T, end));
  cut[(size_t)T] = hi;
  run_parallel(T, [&](int k) {
    bed_index_chunk(f.data, cut[(size_t)k], std::min(cut[(size_t)k + 1], hi), end, piece_rows, found[(size_t)k], errors[(size_t)k], path);
  });
  for (auto& e : errors)
    if (!e.empty()) {
      set_error("%s", e.c_str());
      return MURAL_E_INVALID;
    }
  int64_t n = 0;
  for (auto& v : found)
    for (auto& pc : v) {
      if (n < cap) {
        MURAL_REQUIRE(names && p_lo && p_hi && p_rows && p_first_start, "NULL output array");
        if ((int64_t)pc.name.size() >= name_cap) {
          set_error("%s: chromosome name longer than %d bytes", path, name_cap - 1);
          return MURAL_E_INVALID;
        }
        std::memcpy(names + n * name_cap, pc.name.data(), pc.name.size());
        names[n * name_cap + (int64_t)pc.name.size()] = '\0';
        p_lo[n] = pc.lo; p_hi[n] = pc.hi; p_rows[n] = pc.rows; p_first_start[n] = pc.first_start;
        if (p_order) {
          p_order[4 * n] = pc.in_order; p_order[4 * n + 1] = pc.last_start; p_order[4 * n + 2] = pc.first_strand; p_order[4 * n + 3] = pc.last_strand;
        }
      }
      ++n;
    }
  *n_pieces = n;
  return MURAL_OK;
}

// Rows skip_rows .. skip_rows + n_rows - 1 of the rows that start in the bytes [byte_lo, byte_hi) (byte_lo = a line start), all of
// chromosome `chrom`: start, end, score (class label), strand (0 '+', 1 '-').  The range is cut at line boundaries and parsed by the
// host threads of mural_bed_read.  A row of another chromosome, a malformed row or fewer rows than asked for: MURAL_E_INVALID (the
// file changed behind the index).
extern "C" int mural_bed_parse_range(const char* path, int64_t byte_lo, int64_t byte_hi, int64_t skip_rows, int64_t n_rows, const char* chrom,
                                     int64_t* start, int64_t* end_, float* score, uint8_t* strand) {
  MURAL_REQUIRE(path && chrom, "NULL argument");
  MURAL_REQUIRE(byte_lo >= 0 && byte_hi >= byte_lo && skip_rows >= 0 && n_rows >= 0, "bad parse range");
  if (n_rows == 0) return MURAL_OK;
  MURAL_REQUIRE(start && end_ && score && strand, "NULL output array");
  MappedFile f;
  if (!f.open(path)) {
    set_error("cannot open BED file %s%s%s", path, f.why == "cannot open" ? "" : ": ", f.why == "cannot open" ? "" : f.why.c_str());
    return MURAL_E_INVALID;
  }
  MURAL_REQUIRE((size_t)byte_hi <= f.size, "%s: parse range behind the end of the file (the file changed behind the index)", path);
  const char* lo = f.data + byte_lo;
  const char* hi = f.data + byte_hi;
  const char* end = f.data + f.size;
  MURAL_REQUIRE(byte_lo == 0 || lo[-1] == '\n', "%s: parse range does not start at a line start", path);
  const size_t span = (size_t)(hi - lo);
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), span / (4u << 20) + 1));
  std::vector<BedChunk> ch((size_t)T);
  const char* p = lo;
  for (int k = 0; k < T; ++k) {
    ch[(size_t)k].lo = p;
    const char* c = k + 1 == T ? hi : std::max(p, std::min(hi, align_line(f.data, lo + span * (size_t)(k + 1) / (size_t)T, end)));
    ch[(size_t)k].hi = c;
    p = c;
  }
  run_parallel(T, [&](int k) { bed_parse_chunk(ch[(size_t)k], path, false, nullptr, nullptr, nullptr, nullptr, nullptr); });
  int64_t n = 0;
  for (auto& c : ch) {
    c.row0 = n;
    n += c.rows;
  }
  if (n < skip_rows + n_rows) {
    set_error("%s: %lld rows in the byte range, the index announced %lld (the file changed behind the index)", path, (long long)n,
              (long long)(skip_rows + n_rows));
    return MURAL_E_INVALID;
  }
  const size_t clen = std::strlen(chrom);
  run_parallel(T, [&](int k) {
    BedChunk& c = ch[(size_t)k];
    if (c.row0 + c.rows <= skip_rows || c.row0 >= skip_rows + n_rows) return;
    const char* q = c.lo;
    int64_t r = c.row0;
    while (q < c.hi && r < skip_rows + n_rows) {
      const char* e = line_end(q, c.hi);
      const char* le = e;
      while (le > q && (le[-1] == '\r' || le[-1] == ' ' || le[-1] == '\t')) --le;
      const size_t len = (size_t)(le - q);
      if (!bed_skip_line(q, len)) {
        if (r >= skip_rows) {
          const char* fld[6];
          size_t flen[6];
          int nf = 0;
          const char* w = q;
          while (w <= le && nf < 6) {
            const char* t = static_cast<const char*>(std::memchr(w, '\t', (size_t)(le - w)));
            if (!t) t = le;
            fld[nf] = w;
            flen[nf] = (size_t)(t - w);
            ++nf;
            w = t + 1;
          }
          long long s = 0, t2 = 0;
          float sc = 0.f;
          char msg[512];
          if (nf < 6) {
            std::snprintf(msg, sizeof(msg), "%s: expected 6 tab-separated BED columns (chrom start end name score strand), got %d at byte offset %lld",
                          path, nf, (long long)(q - f.data));
            c.error = msg;
            return;
          }
          const bool ok = parse_i64_field(fld[1], flen[1], &s) && parse_i64_field(fld[2], flen[2], &t2) &&
                          parse_score_field(fld[4], flen[4], &sc) && flen[5] == 1 && (fld[5][0] == '+' || fld[5][0] == '-');
          if (!ok || s < 0 || t2 < s) {
            std::snprintf(msg, sizeof(msg), "%s: malformed BED row at byte offset %lld", path, (long long)(q - f.data));
            c.error = msg;
            return;
          }
          if (flen[0] != clen || std::memcmp(fld[0], chrom, clen) != 0) {
            std::snprintf(msg, sizeof(msg), "%s: row at byte offset %lld is not on chromosome %s (the file changed behind the index)", path,
                          (long long)(q - f.data), chrom);
            c.error = msg;
            return;
          }
          const int64_t o = r - skip_rows;
          start[o] = s;
          end_[o] = t2;
          score[o] = sc;
          strand[o] = fld[5][0] == '-' ? 1 : 0;
        }
        ++r;
      }
      q = e < c.hi ? e + 1 : c.hi;
    }
  });
  for (auto& c : ch)
    if (!c.error.empty()) {
      set_error("%s", c.error.c_str());
      return MURAL_E_INVALID;
    }
  return MURAL_OK;
}

// ---- host twin of mural_op_dense_to_symbols ------------------------------------------------------------------------------------------
// The reference's loader yields HOST tensors (y, cont_x, cat_x, distal_x) and its predict loop moves them to the device batch by batch
// (MuRaL/model/nn_utils.py:37-76; 16 rows per batch by default, commands/predict.py:90): 32 KB of fp32 one-hot per site over PCIe.
// model_predict_m classifies the windows on the host instead -- one symbol byte per column (MURAL_SYM_*, dense_symbol.h's rule) -- and
// uploads 1 / 16 of the bytes.  xs[b]: HOST float [rows[b]][4][L] contiguous; sym: HOST uint8 [sum rows][L] (rows of the batches side
// by side); *n_bad: columns that are no MuRaL encoding (they get code 255; the caller raises).  The batches are spread over the host
// threads (MURAL_HOST_THREADS; up to 64 here: the pass streams 16 KB per output row).  The column loop is host_classify.cpp's.
namespace mural { __attribute__((visibility("hidden"))) int64_t classify_window_host(const float* x, int L, uint8_t* out, const uint8_t* lut625, const uint8_t* lut16); }
namespace {

struct SymLut {
  uint8_t t[625], t16[16];
  SymLut() {
    std::memset(t, 255, sizeof(t));
    const int key[15] = {1, 5, 25, 125, 468, 52, 260, 12, 60, 252, 300, 620, 604, 524, 124};      // dense_symbol.h: d0 + 5 d1 + 25 d2 + 125 d3
    for (int i = 0; i < 15; ++i) t[key[i]] = (uint8_t)i;
    for (int i = 0; i < 16; ++i) t16[i] = t[(i & 1) + 5 * ((i >> 1) & 1) + 25 * ((i >> 2) & 1) + 125 * ((i >> 3) & 1)];      // columns of 0 / 1 only
  }
};
const SymLut kSymLut;

}  // namespace

extern "C" int mural_host_dense_to_symbols(const float* const* xs, const int64_t* rows, int64_t n_batches, int32_t L, uint8_t* sym,
                                           int64_t* n_bad) {
  MURAL_REQUIRE(n_batches >= 0 && L >= 1 && n_bad, "host dense_to_symbols: bad arguments");
  *n_bad = 0;
  if (n_batches == 0) return MURAL_OK;
  MURAL_REQUIRE(xs && rows && sym, "host dense_to_symbols: NULL argument");
  std::vector<int64_t> first((size_t)n_batches + 1, 0);
  for (int64_t b = 0; b < n_batches; ++b) {
    MURAL_REQUIRE(rows[b] >= 0 && (rows[b] == 0 || xs[b]), "host dense_to_symbols: batch %lld is NULL", (long long)b);
    first[(size_t)b + 1] = first[(size_t)b] + rows[b];
  }
  const int64_t total = first[(size_t)n_batches];
  int T = host_threads();
  if (!std::getenv("MURAL_HOST_THREADS")) T = (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency()));
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, total / 256 + 1));
  std::vector<int64_t> bad((size_t)T, 0);
  run_parallel(T, [&](int k) {
    // rows [lo, hi) of the concatenation: whole windows, whichever batch they sit in
    const int64_t lo = total * k / T, hi = total * (k + 1) / T;
    int64_t b = (int64_t)(std::upper_bound(first.begin(), first.end(), lo) - first.begin()) - 1;
    int64_t n = 0;
    for (int64_t r = lo; r < hi; ++r) {
      while (r >= first[(size_t)b + 1]) ++b;
      n += classify_window_host(xs[b] + (size_t)(r - first[(size_t)b]) * 4 * (size_t)L, L, sym + (size_t)r * (size_t)L, kSymLut.t, kSymLut.t16);
    }
    bad[(size_t)k] = n;
  });
  for (int64_t v : bad) *n_bad += v;
  return MURAL_OK;
}

// the small fields of the same batches (y, cat_x) side by side in one staging buffer: bytes[b] bytes from srcs[b], in order
extern "C" int mural_host_concat(const void* const* srcs, const int64_t* bytes, int64_t n, void* dst) {
  MURAL_REQUIRE(n >= 0 && (n == 0 || (srcs && bytes && dst)), "host concat: bad arguments");
  char* o = static_cast<char*>(dst);
  for (int64_t b = 0; b < n; ++b) {
    MURAL_REQUIRE(bytes[b] >= 0 && (bytes[b] == 0 || srcs[b]), "host concat: piece %lld is NULL", (long long)b);
    std::memcpy(o, srcs[b], (size_t)bytes[b]);
    o += bytes[b];
  }
  return MURAL_OK;
}

// Row order of bed_reader (preprocessing.py:39-106) for rows in file order: sites are cut into central_bp-wide segments
// along each chromosome (the first chromosome's grid starts at its first site, later ones at 1), and every segment
// yields its '+' rows, then its '-' rows.  order[k] = input row of output row k; group[k] = index of the yielded group.
extern "C" int mural_bed_segment_order(const int32_t* chrom_id, const int64_t* start, const uint8_t* strand, int64_t n,
                                       int64_t central_bp, int64_t* order, int64_t* group, int64_t* n_groups) {
  MURAL_REQUIRE(n == 0 || (chrom_id && start && strand && order), "NULL argument");
  MURAL_REQUIRE(central_bp >= 1, "central_bp must be positive");
  std::vector<int64_t> pos_rows, neg_rows;
  int64_t out = 0, g = 0;
  auto flush = [&]() {
    if (!pos_rows.empty()) {
      for (int64_t r : pos_rows) {
        order[out] = r;
        if (group) group[out] = g;
        ++out;
      }
      ++g;
      pos_rows.clear();
    }
    if (!neg_rows.empty()) {
      for (int64_t r : neg_rows) {
        order[out] = r;
        if (group) group[out] = g;
        ++out;
      }
      ++g;
      neg_rows.clear();
    }
  };
  int32_t cur = -1;
  int64_t end0 = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (i == 0) {
      cur = chrom_id[0];
      end0 = start[0] + central_bp;
    }
    if (chrom_id[i] != cur) {
      flush();
      cur = chrom_id[i];
      end0 = 1 + central_bp;
    }
    if (start[i] > end0) {
      flush();
      while (start[i] > end0) end0 += central_bp;
    }
    (strand[i] ? neg_rows : pos_rows).push_back(i);
  }
  flush();
  if (n_groups) *n_groups = g;
  return MURAL_OK;
}
