// The HR2 leg (hydra_api/hydra_cpu.cpp:4-127) on the HIP core, driven the way hydra_api's CommandBuffer::CommitToStorage + hr2CommitAndRender drive
// the reference's driver:  hydra_hip_hr2 <scene.xml> <width> <height> <spp> <out.bin>
//   1. the client reads every mesh of the scene into ITS OWN arrays (here: from the scene's .vsgf files; an HR2 client builds them with the mesh
//      API) and marks the mesh nodes ptrs="1" in the scene description it keeps in memory;
//   2. driver->LoadScene(description, {mesh id -> pointers}, update flags)  - no geometry file is opened by the driver;
//   3. driver->CommitDeviceData();  driver->Render(0, 0, W, H, 4, frame, spp)  = SetFrameBufferSize, SetViewport, UpdateMembersPlainData, PackXYBlock,
//      PathTraceBlock.
// A second LoadScene without SCN_UPDATE_GEOMETRY keeps the mesh pointers of the first (hydra_cpu.cpp:34). The test compares the frame with
// hydra_hip_render's (the file path of the same scene): bit-identical.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../hydracore3_amd/csrc/hydra_driver_hip.h"

using namespace hydra_hip;

struct ClientMesh { std::vector<float> pos4, norm4, tang4, uv2; std::vector<uint32_t> idx, mat; };

static bool readVsgf(const std::string& path, ClientMesh& m)
{
  std::vector<uint8_t> f; if (!detail::readFile(path, f) || f.size() < 24) return false;
  uint32_t nv, ni, nm, flags; std::memcpy(&nv, f.data() + 8, 4); std::memcpy(&ni, f.data() + 12, 4); std::memcpy(&nm, f.data() + 16, 4); std::memcpy(&flags, f.data() + 20, 4);
  size_t off = 24;
  m.pos4.assign((const float*)(f.data() + off), (const float*)(f.data() + off) + (size_t)nv * 4); off += (size_t)nv * 16;
  m.norm4.assign((size_t)nv * 4, 0.0f);
  if (!(flags & 8u)) { std::memcpy(m.norm4.data(), f.data() + off, (size_t)nv * 16); off += (size_t)nv * 16; }
  if (flags & 1u) { m.tang4.assign((const float*)(f.data() + off), (const float*)(f.data() + off) + (size_t)nv * 4); off += (size_t)nv * 16; }
  m.uv2.assign((const float*)(f.data() + off), (const float*)(f.data() + off) + (size_t)nv * 2); off += (size_t)nv * 8;
  m.idx.assign((const uint32_t*)(f.data() + off), (const uint32_t*)(f.data() + off) + ni); off += (size_t)ni * 4;
  m.mat.assign((const uint32_t*)(f.data() + off), (const uint32_t*)(f.data() + off) + ni / 3);
  return true;
}

int main(int argc, char** argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: %s <scene.xml> <width> <height> <spp> <out.bin>\n", argv[0]); return 2; }
  const std::string xmlPath = argv[1];
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), spp = std::atoi(argv[4]);
  std::vector<uint8_t> raw; if (!detail::readFile(xmlPath, raw)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  std::string text(raw.begin(), raw.end());
  const size_t slash = xmlPath.find_last_of("/\\");
  const std::string folder = slash == std::string::npos ? std::string(".") : xmlPath.substr(0, slash);

  // 1. the client's meshes: parsed once to find the mesh nodes, read into client memory, nodes marked ptrs="1"
  XmlNode root; std::string err; { XmlParser parser(text); if (!parser.parse(root, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; } }
  std::vector<ClientMesh> meshes; std::unordered_map<int, HR2::MeshPointers> meshPtrById;
  if (const XmlNode* lib = root.child("geometry_lib")) {
    const auto nodes = lib->all("mesh");
    meshes.resize(nodes.size());
    for (size_t k = 0; k < nodes.size(); k++) {
      if (!readVsgf(folder + "/" + nodes[k]->get("loc"), meshes[k])) { std::fprintf(stderr, "cannot read mesh %s\n", nodes[k]->get("loc").c_str()); return 1; }
      HR2::MeshPointers mp; const ClientMesh& m = meshes[k];
      mp.vPosPtr = m.pos4.data(); mp.vPosStride = 4; mp.vNormPtr = m.norm4.data(); mp.vTangPtr = m.tang4.empty() ? nullptr : m.tang4.data(); mp.vTexCoordPtr = m.uv2.data();
      mp.vertNum = (uint32_t)(m.pos4.size() / 4); mp.indicesPtr = m.idx.data(); mp.indicesNum = (uint32_t)m.idx.size();
      mp.matIdPtr = m.mat.data(); mp.matIdNum = (uint32_t)m.mat.size(); mp.matIdAll = m.mat.empty() ? 0u : m.mat[0];
      meshPtrById[std::atoi(nodes[k]->get("id").c_str())] = mp;
    }
  }
  // mark the nodes in the description: <mesh id=".." ...> -> <mesh ptrs="1" id=".." ...> and drop nothing else (loc stays, unread)
  for (size_t p = text.find("<mesh "); p != std::string::npos; p = text.find("<mesh ", p + 6)) text.insert(p + 6, "ptrs=\"1\" ");

  // 2. / 3. the driver
  auto driver = std::make_shared<HR2::HydraHipRenderDriver>(0);
  if (!driver->valid()) { std::fprintf(stderr, "[hydra_hip_hr2]: no HIP device\n"); return 1; }
  // (width / height of the description are overridden by Render's size, as in hydra_api; the loader needs them > 0 to lay the camera out)
  HR2::RDScene_Input input; input.pMeshPtrs = &meshPtrById;
  if (!driver->LoadScene(text, folder, input, HR2::SCN_UPDATE_ALL)) { std::fprintf(stderr, "[hydra_hip_hr2]: %s\n", driver->lastError().c_str()); return 1; }
  // a second commit that does not touch geometry: the driver keeps the pointers it has (a client would have changed a material or the camera)
  HR2::RDScene_Input none;
  if (!driver->LoadScene(text, folder, none, HR2::SCN_UPDATE_ALL & ~HR2::SCN_UPDATE_GEOMETRY)) { std::fprintf(stderr, "[hydra_hip_hr2]: %s\n", driver->lastError().c_str()); return 1; }
  driver->CommitDeviceData();
  std::vector<float> frame((size_t)W * H * 4, 0.0f);
  driver->Render(0, 0, W, (uint32_t)H, 4, frame.data(), (uint32_t)spp);
  if (!driver->lastError().empty()) { std::fprintf(stderr, "[hydra_hip_hr2]: %s\n", driver->lastError().c_str()); return 1; }
  FILE* f = std::fopen(argv[5], "wb"); if (!f) return 1;
  std::fwrite(frame.data(), sizeof(float), frame.size(), f); std::fclose(f);
  double s = 0.0; for (size_t i = 0; i < frame.size(); i += 4) s += frame[i] + frame[i + 1] + frame[i + 2];
  std::printf("[hydra_hip_hr2]: %zu meshes by pointers, %dx%d @ %d spp, mean radiance %.5f\n", meshes.size(), W, H, spp, s / (3.0 * W * H * spp));
  return 0;
}
