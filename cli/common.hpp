// cli/common.hpp -- shared pieces of the drop-in command line tools (hmt/main_merge_order_pb.cxx,
// hmt/main_merge_order_bc.cxx).  GLIA reads images through ITK, which this image does not have; the tools here
// read MetaImage files (.mha / .mhd + raw, plain or zlib-compressed -- what ITK writes natively --) and keep GLIA's flags,
// output formats (util/text_io.hxx:103-133) and error behaviour (message on stderr, exit status 1).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <zlib.h>
#include <set>
#include <sstream>
#include <unistd.h>
#include <string>
#include <vector>

#include "../include/glia_hmt.h"
#include "text_io.hpp"      // perr, writeOrder / writeDoubles / writeRows: util/text_io.hxx:103-133 (host-only header, also built into cli/text_io_check)

namespace cli {

inline void check(int rc) { if (rc) perr(glia_hmt_last_error()); }
inline void hipCheck(hipError_t e) { if (e != hipSuccess) perr(std::string("Error: HIP: ") + hipGetErrorString(e)); }

struct Volume {
  int dim = 0;
  int64_t dims[3] = {1, 1, 1};
  int64_t full_nz = 1;             // planes of the FILE (dims[2] = planes read, when only a z range was asked for)
  std::vector<uint32_t> u32;
  std::vector<float> f32;
  size_t size() const { return (size_t)dims[0] * dims[1] * dims[2]; }
};

// MetaImage reader: ObjectType = Image, NDims 2|3, ElementType MET_{UCHAR,USHORT,UINT,ULONG,SHORT,INT,FLOAT,DOUBLE},
// CompressedData = False | True (one zlib stream, itk::MetaImageIO), ElementDataFile = LOCAL | <file>
// zFirst / zCount (3D only, zCount >= 0): read only those planes -- a rank of the slab route reads its own z range
inline Volume readMetaImage(const std::string& file, bool wantFloat, int64_t zFirst = 0, int64_t zCount = -1) {
  std::ifstream is(file, std::ios::binary);
  if (!is) perr("Error: cannot open file " + file);
  std::map<std::string, std::string> kv;
  std::string line;
  std::streampos dataPos = 0;
  while (std::getline(is, line)) {
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    auto trim = [](std::string s) { size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
    std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
    kv[k] = v;
    if (k == "ElementDataFile") { dataPos = is.tellg(); break; }
  }
  const bool compressed = kv.count("CompressedData") && (kv["CompressedData"] == "True" || kv["CompressedData"] == "true");
  Volume vol;
  vol.dim = atoi(kv["NDims"].c_str());
  if (vol.dim != 2 && vol.dim != 3) perr("Error: unsupported image dimension in " + file);
  std::istringstream ds(kv["DimSize"]);
  for (int i = 0; i < vol.dim; ++i) ds >> vol.dims[i];
  const std::string et = kv["ElementType"];
  size_t es = et == "MET_UCHAR" ? 1 : (et == "MET_USHORT" || et == "MET_SHORT") ? 2 : (et == "MET_UINT" || et == "MET_INT" || et == "MET_FLOAT") ? 4
              : (et == "MET_ULONG" || et == "MET_DOUBLE" || et == "MET_ULONG_LONG") ? 8 : 0;
  if (!es) perr("Error: unsupported MetaImage element type " + et);
  vol.full_nz = vol.dims[2];
  std::streamoff skip = 0;
  if (zCount >= 0) {
    if (vol.dim != 3 || zFirst < 0 || zFirst + zCount > vol.dims[2]) perr("Error: plane range outside the image " + file);
    skip = (std::streamoff)((size_t)zFirst * vol.dims[0] * vol.dims[1] * es);
    vol.dims[2] = zCount;
  }
  const size_t n = vol.size();
  std::vector<char> raw(n * es);
  const std::string dir = file.substr(0, file.find_last_of('/') == std::string::npos ? 0 : file.find_last_of('/') + 1);
  if (compressed) {
    // one zlib stream over the whole image: inflate it, keep the bytes of the wanted planes
    std::vector<char> packed;
    if (kv["ElementDataFile"] == "LOCAL") { is.seekg(dataPos); packed.assign(std::istreambuf_iterator<char>(is), std::istreambuf_iterator<char>()); }
    else {
      std::ifstream rs(dir + kv["ElementDataFile"], std::ios::binary);
      if (!rs) perr("Error: cannot open file " + dir + kv["ElementDataFile"]);
      packed.assign(std::istreambuf_iterator<char>(rs), std::istreambuf_iterator<char>());
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) perr("Error: zlib initialisation failed");
    std::vector<char> chunk(1 << 22);
    size_t inPos = 0, outPos = 0;                 // outPos: bytes of the whole image produced so far
    const size_t want0 = (size_t)skip, want1 = (size_t)skip + raw.size();
    int zr = Z_OK;
    while (zr != Z_STREAM_END && outPos < want1) {
      if (zs.avail_in == 0 && inPos < packed.size()) {
        const size_t take = std::min<size_t>(packed.size() - inPos, (size_t)1 << 30);
        zs.next_in = (Bytef*)(packed.data() + inPos); zs.avail_in = (uInt)take; inPos += take;
      }
      zs.next_out = (Bytef*)chunk.data(); zs.avail_out = (uInt)chunk.size();
      zr = inflate(&zs, Z_NO_FLUSH);
      if (zr != Z_OK && zr != Z_STREAM_END) { inflateEnd(&zs); perr("Error: corrupt compressed image data in " + file); }
      const size_t got = chunk.size() - zs.avail_out;
      const size_t a = std::max(outPos, want0), b = std::min(outPos + got, want1);
      if (a < b) memcpy(raw.data() + (a - want0), chunk.data() + (a - outPos), b - a);
      outPos += got;
      if (got == 0 && zs.avail_in == 0 && inPos >= packed.size()) break;
    }
    inflateEnd(&zs);
    if (outPos < want1) perr("Error: truncated image data in " + file);
  } else if (kv["ElementDataFile"] == "LOCAL") { is.seekg(dataPos + skip); is.read(raw.data(), raw.size()); if (!is) perr("Error: truncated image data in " + file); }
  else {
    std::ifstream rs(dir + kv["ElementDataFile"], std::ios::binary);
    if (!rs) perr("Error: cannot open file " + dir + kv["ElementDataFile"]);
    rs.seekg(skip);
    rs.read(raw.data(), raw.size());
    if (!rs) perr("Error: truncated image data in " + file);
  }
  auto get = [&](size_t i) -> double {
    const char* p = raw.data() + i * es;
    if (et == "MET_UCHAR") return *(const uint8_t*)p;
    if (et == "MET_USHORT") return *(const uint16_t*)p;
    if (et == "MET_SHORT") return *(const int16_t*)p;
    if (et == "MET_UINT") return *(const uint32_t*)p;
    if (et == "MET_INT") return *(const int32_t*)p;
    if (et == "MET_FLOAT") return *(const float*)p;
    if (et == "MET_DOUBLE") return *(const double*)p;
    return (double)*(const uint64_t*)p;
  };
  if (wantFloat) { vol.f32.resize(n); for (size_t i = 0; i < n; ++i) vol.f32[i] = (float)get(i); }
  else { vol.u32.resize(n); for (size_t i = 0; i < n; ++i) vol.u32[i] = (uint32_t)get(i); }
  return vol;
}

template <typename T> T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  hipCheck(hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())));
  hipCheck(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return d;
}

// boost::program_options-like parsing: --long value | -s value | multitoken | repeated options accumulate
struct Args {
  std::map<std::string, std::vector<std::string>> v;
  bool has(const std::string& k) const { return v.count(k) > 0; }
  std::string str(const std::string& k, const std::string& def = "") const { auto it = v.find(k); return it == v.end() || it->second.empty() ? def : it->second.back(); }
  std::vector<std::string> all(const std::string& k) const { auto it = v.find(k); return it == v.end() ? std::vector<std::string>() : it->second; }
};
inline Args parse(int argc, char** argv, const std::map<std::string, std::string>& shortToLong, const std::vector<std::string>& known,
                  const std::string& usage) {
  Args a;
  std::string cur;
  for (int i = 1; i < argc; ++i) {
    std::string t = argv[i];
    bool isOpt = t.size() > 1 && t[0] == '-' && !(isdigit((unsigned char)t[1]) || t[1] == '.');
    if (isOpt) {
      std::string name = t[1] == '-' ? t.substr(2) : (shortToLong.count(t.substr(1)) ? shortToLong.at(t.substr(1)) : "?");
      size_t eq = name.find('=');
      std::string val;
      if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); }
      bool ok = false;
      for (auto& k : known) if (k == name) ok = true;
      if (name == "help") { std::cerr << usage << std::endl; exit(EXIT_FAILURE); }
      if (!ok) { std::cerr << "Error: unrecognised option '" << t << "'" << std::endl << usage << std::endl; exit(EXIT_FAILURE); }
      cur = name;
      a.v[cur];
      if (eq != std::string::npos) a.v[cur].push_back(val);
    } else {
      if (cur.empty()) { std::cerr << "Error: too many positional options" << std::endl << usage << std::endl; exit(EXIT_FAILURE); }
      a.v[cur].push_back(t);
    }
  }
  return a;
}

// MetaImage writer (single .mha file); 16-bit output for --write16 (castWriteImage<UInt16Image>); compress = the tools'
// --compress / -z (itk::ImageFileWriter::SetUseCompression, util/image_io.hxx:46-52): one zlib stream, CompressedDataSize in the header
inline void writeMetaImageBytes(const std::string& file, int dim, const int64_t dims[3], const char* elementType, const void* data, size_t bytes, bool compress) {
  std::ofstream os(file, std::ios::binary);
  if (!os) perr("Error: cannot create file " + file);
  std::vector<char> packed;
  if (compress) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit(&zs, 2) != Z_OK) perr("Error: zlib initialisation failed");
    std::vector<char> chunk(1 << 22);
    size_t inPos = 0;
    int zr = Z_OK;
    while (zr != Z_STREAM_END) {
      if (zs.avail_in == 0 && inPos < bytes) {
        const size_t take = std::min<size_t>(bytes - inPos, (size_t)1 << 30);
        zs.next_in = (Bytef*)((const char*)data + inPos); zs.avail_in = (uInt)take; inPos += take;
      }
      zs.next_out = (Bytef*)chunk.data(); zs.avail_out = (uInt)chunk.size();
      zr = deflate(&zs, inPos >= bytes ? Z_FINISH : Z_NO_FLUSH);
      if (zr == Z_STREAM_ERROR) { deflateEnd(&zs); perr("Error: compression failed for " + file); }
      packed.insert(packed.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
    }
    deflateEnd(&zs);
  }
  os << "ObjectType = Image\nNDims = " << dim << "\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = " << (compress ? "True" : "False");
  if (compress) os << "\nCompressedDataSize = " << packed.size();
  os << "\nDimSize =";
  for (int i = 0; i < dim; ++i) os << " " << dims[i];
  os << "\nElementType = " << elementType << "\nElementDataFile = LOCAL\n";
  if (compress) os.write(packed.data(), packed.size()); else os.write((const char*)data, bytes);
}
inline void writeMetaImage(const std::string& file, int dim, const int64_t dims[3], const std::vector<uint32_t>& v, bool as16, bool compress = false) {
  if (as16) {
    std::vector<uint16_t> w(v.size());
    for (size_t i = 0; i < v.size(); ++i) w[i] = (uint16_t)v[i];
    writeMetaImageBytes(file, dim, dims, "MET_USHORT", w.data(), w.size() * 2, compress);
  } else writeMetaImageBytes(file, dim, dims, "MET_UINT", v.data(), v.size() * 4, compress);
}

inline void writeMetaImageFloat(const std::string& file, int dim, const int64_t dims[3], const std::vector<float>& v, bool compress = false) {
  writeMetaImageBytes(file, dim, dims, "MET_FLOAT", v.data(), v.size() * 4, compress);
}

// readData(order, file, true) of util/text_io.hxx:193-213 for TTriple<Label> (type/tuple.hxx:26-28)
inline std::vector<uint32_t> readOrder(const std::string& file) {
  std::ifstream is(file);
  if (!is) perr("Error: invalid data file dimension in " + file);
  std::vector<uint32_t> o;
  unsigned long long x;
  while (is >> x) o.push_back((uint32_t)x);
  if (o.size() % 3) perr("Error: invalid data file dimension in " + file);
  return o;
}

// optional mask image (-m): uploaded as a u32 volume, NULL when the option is absent
inline uint32_t* loadMask(const Args& a, const char* key, size_t expect) {
  if (!a.has(key)) return nullptr;
  Volume m = readMetaImage(a.str(key), false);
  if (m.size() != expect) perr("Error: image sizes do not match...");
  return upload(m.u32);
}

// --slabs N [--rank r --commId file [--device d]]: the context and the communicator of the slab route (include/glia_hmt.h).  Without
// --rank all N ranks live in this process; with it this process is rank r of N over RCCL and rank 0 hands RCCL's unique id to the
// others through the file.
inline glia_hmt_comm* makeSlabComm(const Args& a, int slabs, glia_hmt_ctx** ctx) {
  const bool multi = a.has("rank");
  const int rank = atoi(a.str("rank", "0").c_str());
  if (multi && (!a.has("commId") || rank < 0 || rank >= slabs)) perr("Error: --rank needs --commId and 0 <= rank < slabs...");
  int ndev = 0;
  hipCheck(hipGetDeviceCount(&ndev));
  const int device = a.has("device") ? atoi(a.str("device").c_str()) : (multi && ndev > 0 ? rank % ndev : 0);
  check(glia_hmt_ctx_create(device, nullptr, ctx));
  glia_hmt_comm* comm = nullptr;
  if (!multi) { check(glia_hmt_comm_create_local(*ctx, slabs, &comm)); return comm; }
  // The id file carries the launch's nonce (--commNonce, default: the parent process id -- one launcher script starts all ranks) in
  // front of RCCL's 128-byte id: a file left over from an earlier launch does not carry it and is never taken for this launch's
  // (a stale id would make ncclCommInitRank hang instead of fail).  Rank 0 removes an old file first, writes the new one atomically
  // (rename) and removes it again once the communicator stands.
  char id[128];
  const std::string f = a.str("commId");
  const unsigned long long nonce = a.has("commNonce") ? strtoull(a.str("commNonce").c_str(), nullptr, 0) : (unsigned long long)getppid();
  if (rank == 0) {
    (void)remove(f.c_str());
    check(glia_hmt_comm_unique_id(id));
    { std::ofstream os(f + ".tmp", std::ios::binary); os.write(reinterpret_cast<const char*>(&nonce), sizeof(nonce)); os.write(id, sizeof(id)); }
    if (rename((f + ".tmp").c_str(), f.c_str())) perr("Error: cannot create file " + f);
  } else {
    for (int tries = 0;; ++tries) {
      std::ifstream is(f, std::ios::binary);
      unsigned long long got = 0;
      if (is && is.read(reinterpret_cast<char*>(&got), sizeof(got)) && got == nonce && is.read(id, sizeof(id))) break;
      if (tries > 1200) perr("Error: cannot open file " + f + " (or it belongs to another launch: --commNonce)");
      usleep(100000);
    }
  }
  check(glia_hmt_comm_create_rccl(*ctx, slabs, rank, id, &comm));
  if (rank == 0) (void)remove(f.c_str());      // (ncclCommInitRank returns when every rank has joined)
  return comm;
}

inline bool flagOf(const Args& a, const char* k) { std::string v = a.str(k, "0"); return v == "1" || v == "true"; }

// prepareImages (hmt/hmt_util.hxx:17-56) + the shape normalisers of hmt/main_merge_order_bc.cxx:36-39 / main_bc_feat.cxx:43-46
struct FeatInputs {
  Volume seg, pb;
  uint32_t* dLab = nullptr;
  float* dPb = nullptr;
  std::map<std::string, float*> volumes;      // every distinct image file is read and uploaded once
  glia_hmt_feat_config cfg;
};
// zFirst / zCount: only those planes of every volume (a slab of the slab route); the shape normalisers stay the whole volume's
inline void loadFeatInputs(const Args& a, FeatInputs& f, int64_t zFirst = 0, int64_t zCount = -1) {
  const std::string pbFile = a.str("pb");
  f.seg = readMetaImage(a.str("segImage"), false, zFirst, zCount);
  f.pb = readMetaImage(pbFile, true, zFirst, zCount);
  if (f.seg.dim != f.pb.dim || f.seg.size() != f.pb.size()) perr("Error: image sizes do not match...");
  f.dLab = upload(f.seg.u32);
  f.dPb = upload(f.pb.f32);
  f.volumes[pbFile] = f.dPb;
  auto volume = [&](const std::string& file) -> float* {
    auto it = f.volumes.find(file);
    if (it != f.volumes.end()) return it->second;
    Volume v = readMetaImage(file, true, zFirst, zCount);
    if (v.size() != f.seg.size()) perr("Error: image sizes do not match...");
    float* d = upload(v.f32);
    f.volumes[file] = d;
    return d;
  };
  memset(&f.cfg, 0, sizeof(f.cfg));
  auto addAll = [&](const char* ki, const char* kb, const char* kl, const char* ku, glia_hmt_image* list, int& n) {
    auto im = a.all(ki), b = a.all(kb), l = a.all(kl), u = a.all(ku);
    for (size_t i = 0; i < im.size(); ++i) {
      if (i >= b.size() || i >= l.size() || i >= u.size()) perr("Error: histogram parameters missing for an input image...");
      if (n >= GLIA_HMT_MAX_IMAGES) perr("Error: too many input images...");
      list[n].d_image = volume(im[i]); list[n].bins = atoi(b[i].c_str()); list[n].lo = atof(l[i].c_str()); list[n].hi = atof(u[i].c_str()); ++n;
    }
  };
  // --rbi images are appended to BOTH the region and the boundary list, before the exclusive ones
  addAll("rbi", "rbb", "rbl", "rbu", f.cfg.region, f.cfg.n_region);
  addAll("rbi", "rbb", "rbl", "rbu", f.cfg.boundary, f.cfg.n_boundary);
  addAll("ri", "rb", "rl", "ru", f.cfg.region, f.cfg.n_region);
  addAll("bi", "bb", "bl", "bu", f.cfg.boundary, f.cfg.n_boundary);
  addAll("rli", "rlb", "rll", "rlu", f.cfg.rlabel, f.cfg.n_rlabel);
  f.cfg.d_pb = f.dPb;
  auto bt = a.all("bt");
  if (bt.size() > GLIA_HMT_MAX_THRESH) perr("Error: too many boundary thresholds for this version...");
  f.cfg.n_thresholds = (int)bt.size();
  for (size_t i = 0; i < bt.size(); ++i) f.cfg.thresholds[i] = atof(bt[i].c_str());
  double vol = 1.0, diag = 0.0;
  for (int i = 0; i < f.seg.dim; ++i) { const double e = (double)(i == 2 ? f.seg.full_nz : f.seg.dims[i]); vol *= e; diag += e * e; }
  f.cfg.normalizing_area = flagOf(a, "ns") ? vol : 1.0;
  f.cfg.normalizing_length = flagOf(a, "ns") ? std::sqrt(diag) : 1.0;
  f.cfg.use_log_shape = flagOf(a, "logs");
  f.cfg.use_simple_features = flagOf(a, "simpf");
  // the reference fixes this layout at build time (cmake -DGLIA_HMT_HIST_FEAT=ON -> GLIA_USE_HISTOGRAM_AS_FEATS); here it is a flag
  f.cfg.use_histogram_features = a.has("histf") ? flagOf(a, "histf") : 0;
  // likewise GLIA_HMT_MEDIAN_FEAT=ON -> GLIA_USE_MEDIAN_AS_FEATS (bc_feat only: the greedy loop refuses it, include/glia_hmt.h)
  f.cfg.use_median_features = a.has("medf") ? flagOf(a, "medf") : 0;
}

}  // namespace cli
