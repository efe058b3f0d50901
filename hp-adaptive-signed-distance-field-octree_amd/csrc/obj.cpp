// Wavefront OBJ reader for the mesh field: Meshing::ObjParser::Load (Source/Meshing/ObjParser.cpp:11-164).
//
// One-off host preprocessing in front of the mesh configs.  The reference reads `v x y z` records
// and triangular `f` records in the three spellings `a b c`, `a//n b//n c//n` and `a/t/n b/t/n c/t/n` (it picks
// the sscanf pattern from which of v / vn / vt records it has seen so far, :82-137); everything else is skipped.
// Here the spelling is read off each token, so the same files give the same arrays and `a/t` tokens work too.
// Indices are 1-based in the file; negative (relative) indices, which the reference does not support, are
// resolved against the vertices read so far.  Faces with more than three corners are rejected (the reference
// would silently keep the first three).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hpsdf.h"

namespace hpsdf {

int loadObj(const char* path, std::vector<float>& verts, std::vector<uint64_t>& tris, std::string& err) {
    std::FILE* fh = std::fopen(path, "rb");
    if (!fh) {
        err = std::string("cannot open ") + path + ": " + std::strerror(errno);
        return HPSDF_ERR_INVALID_ARGUMENT;
    }
    std::vector<char> data;
    {
        std::fseek(fh, 0, SEEK_END);
        const long sz = std::ftell(fh);
        std::fseek(fh, 0, SEEK_SET);
        data.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = data.empty() ? 0 : std::fread(data.data(), 1, data.size(), fh);
        std::fclose(fh);
        data.resize(got);
        for (char& c : data)
            if (c == '\0') c = ' ';  // (a NUL inside the file would end the text for strchr below, in front of the line's newline)
        data.push_back('\n');
        data.push_back('\0');
    }
    verts.clear();
    tris.clear();
    size_t lineNo = 0;
    for (char* p = data.data(); *p;) {
        char* eol = std::strchr(p, '\n');
        *eol = '\0';
        ++lineNo;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* q = p + 1;
            float v[3];
            for (int a = 0; a < 3; ++a) {
                char* e = nullptr;
                v[a] = std::strtof(q, &e);
                if (e == q) {
                    err = "malformed vertex on line " + std::to_string(lineNo);
                    return HPSDF_ERR_INVALID_ARGUMENT;
                }
                q = e;
            }
            verts.insert(verts.end(), v, v + 3);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            char* q = p + 1;
            int corners = 0;
            uint64_t idx[3];
            for (;;) {
                while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
                if (!*q) break;
                char* e = nullptr;
                const long long i = std::strtoll(q, &e, 10);
                if (e == q) {
                    err = "malformed face on line " + std::to_string(lineNo);
                    return HPSDF_ERR_INVALID_ARGUMENT;
                }
                if (corners == 3) {
                    err = "only triangles are supported (line " + std::to_string(lineNo) + ")";
                    return HPSDF_ERR_UNSUPPORTED;
                }
                const long long nv = (long long)(verts.size() / 3);
                const long long z = i > 0 ? i - 1 : nv + i;  // 1-based, or relative to the vertices so far
                if (i == 0 || z < 0) {
                    err = "vertex index out of range on line " + std::to_string(lineNo);
                    return HPSDF_ERR_INVALID_ARGUMENT;
                }
                idx[corners++] = (uint64_t)z;
                q = e;
                while (*q && *q != ' ' && *q != '\t' && *q != '\r') ++q;  // skip /t/n
            }
            if (corners != 3) {
                err = "face with fewer than three corners on line " + std::to_string(lineNo);
                return HPSDF_ERR_INVALID_ARGUMENT;
            }
            tris.insert(tris.end(), idx, idx + 3);
        }
        p = eol + 1;
    }
    const uint64_t nv = verts.size() / 3;
    for (uint64_t t : tris)
        if (t >= nv) {
            err = "face references a vertex that does not exist";
            return HPSDF_ERR_INVALID_ARGUMENT;
        }
    if (verts.empty() || tris.empty()) {  // ObjParser::Load returns false (:34)
        err = "no vertices or no triangles in " + std::string(path);
        return HPSDF_ERR_INVALID_ARGUMENT;
    }
    return HPSDF_OK;
}

}  // namespace hpsdf
