// ggml-model.bin reader -- see model_file.h.  Failure cases follow biogpt.cpp:34-48 (open/magic),
// :76-80 (vocab count), :160-165 (ftype), :394-417 (unknown tensor / wrong shape / wrong size),
// :442-447 (tensor count).
#include "model_file.h"

#include <cerrno>
#include <cstdlib>

namespace bg {

namespace {
thread_local std::string g_err;

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

bool read_i32(FILE *f, int32_t &v) { return fread(&v, 4, 1, f) == 1; }

bool read_strings(FILE *f, int32_t count, std::vector<std::string> &out) {
    out.resize((size_t)count);
    std::vector<char> buf;
    for (int32_t i = 0; i < count; i++) {
        uint32_t len;
        if (fread(&len, 4, 1, f) != 1) return false;
        if (len > (1u << 20)) return false;
        buf.resize(len);
        if (len && fread(buf.data(), 1, len, f) != len) return false;
        out[(size_t)i].assign(buf.data(), len);
    }
    return true;
}
}  // namespace

void set_error(const char *func, const char *fmt, ...) {
    char msg[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    g_err = std::string(func) + ": " + msg;
    fprintf(stderr, "%s\n", g_err.c_str());
}
const char *last_error() { return g_err.c_str(); }
void clear_error() { g_err.clear(); }

std::vector<ExpectedTensor> expected_tensors(const biogpt_hip_hparams &hp) {
    const int64_t D = hp.d_model, F = hp.d_ff, V = hp.n_vocab;
    std::vector<ExpectedTensor> v;
    v.push_back({"output_projection.weight", D, V, true});
    v.push_back({"biogpt.embed_tokens.weight", D, V, true});
    v.push_back({"biogpt.embed_positions.weight", D, -1, true});
    v.push_back({"biogpt.layer_norm.weight", D, 1, false});
    v.push_back({"biogpt.layer_norm.bias", D, 1, false});
    for (int i = 0; i < hp.n_layer; i++) {
        const std::string p = "biogpt.layers." + std::to_string(i) + ".";
        for (const char *proj : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
            v.push_back({p + "self_attn." + proj + ".weight", D, D, true});
            v.push_back({p + "self_attn." + proj + ".bias", D, 1, false});
        }
        v.push_back({p + "self_attn_layer_norm.weight", D, 1, false});
        v.push_back({p + "self_attn_layer_norm.bias", D, 1, false});
        v.push_back({p + "final_layer_norm.weight", D, 1, false});
        v.push_back({p + "final_layer_norm.bias", D, 1, false});
        v.push_back({p + "fc1.weight", D, F, true});
        v.push_back({p + "fc1.bias", F, 1, false});
        v.push_back({p + "fc2.weight", F, D, true});
        v.push_back({p + "fc2.bias", D, 1, false});
    }
    return v;
}

const TensorEntry *ModelFile::find(const std::string &name) const {
    auto it = by_name.find(name);
    return it == by_name.end() ? nullptr : &tensors[(size_t)it->second];
}

bool ModelFile::open(const std::string &fname) {
    path = fname;
    File fh;
    fh.f = fopen(fname.c_str(), "rb");
    if (!fh.f) BG_FAIL(false, "failed to open '%s'", fname.c_str());
    FILE *f = fh.f;

    uint32_t magic = 0;
    if (fread(&magic, 4, 1, f) != 1 || magic != FILE_MAGIC)
        BG_FAIL(false, "invalid model file '%s' (bad magic)", fname.c_str());

    int32_t h[7];
    for (int i = 0; i < 7; i++)
        if (!read_i32(f, h[i])) BG_FAIL(false, "invalid model file '%s' (truncated header)", fname.c_str());
    hp.n_vocab = h[0]; hp.n_layer = h[1]; hp.n_head = h[2]; hp.n_positions = h[3];
    hp.d_ff = h[4]; hp.d_model = h[5]; hp.ftype = h[6];
    if (hp.n_vocab <= 0 || hp.n_layer < 0 || hp.n_head <= 0 || hp.n_positions <= 0 || hp.d_ff <= 0 ||
        hp.d_model <= 0 || hp.d_model % hp.n_head != 0)
        BG_FAIL(false, "invalid model file '%s' (bad hyperparameters)", fname.c_str());

    int32_t n_vocab = 0;
    if (!read_i32(f, n_vocab) || n_vocab != hp.n_vocab)
        BG_FAIL(false, "invalid model file '%s' (bad vocab size %d != %d)", fname.c_str(), n_vocab, hp.n_vocab);
    if (!read_strings(f, n_vocab, vocab)) BG_FAIL(false, "invalid model file '%s' (truncated vocab)", fname.c_str());

    int32_t n_merges = 0;
    if (!read_i32(f, n_merges) || n_merges < 0)
        BG_FAIL(false, "invalid model file '%s' (bad merge size %d)", fname.c_str(), n_merges);
    if (!read_strings(f, n_merges, merges)) BG_FAIL(false, "invalid model file '%s' (truncated merges)", fname.c_str());
    hp.n_merges = n_merges;  // F6: the reference insists on 40000; the count in the file is what counts here

    if (ftype_to_type(hp.ftype) == T_INVALID)
        BG_FAIL(false, "invalid model file '%s' (bad ftype value %d)", fname.c_str(), hp.ftype);

    // tensor directory (biogpt.cpp:369-434)
    for (;;) {
        int32_t n_dims, length, ttype;
        if (!read_i32(f, n_dims)) break;  // EOF
        if (!read_i32(f, length) || !read_i32(f, ttype))
            BG_FAIL(false, "invalid model file '%s' (truncated tensor header)", fname.c_str());
        if (n_dims < 1 || n_dims > 2 || length <= 0 || length > 255)
            BG_FAIL(false, "invalid model file '%s' (bad tensor header)", fname.c_str());
        TensorEntry t;
        t.n_dims = n_dims;
        t.type = ttype;
        int32_t ne[2] = {1, 1};
        for (int i = 0; i < n_dims; i++)
            if (!read_i32(f, ne[i]) || ne[i] <= 0)
                BG_FAIL(false, "invalid model file '%s' (bad tensor dims)", fname.c_str());
        t.ne0 = ne[0];
        t.ne1 = ne[1];
        std::vector<char> nm((size_t)length);
        if (fread(nm.data(), 1, (size_t)length, f) != (size_t)length)
            BG_FAIL(false, "invalid model file '%s' (truncated tensor name)", fname.c_str());
        t.name.assign(nm.data(), (size_t)length);
        if (t.type != T_F32 && t.type != T_F16 && !is_quantized(t.type))
            BG_FAIL(false, "tensor '%s' has unsupported type %d", t.name.c_str(), t.type);
        if (is_quantized(t.type) && (t.ne0 % QK) != 0)
            BG_FAIL(false, "tensor '%s' has wrong size in model file (row %lld not a multiple of %d)",
                    t.name.c_str(), (long long)t.ne0, QK);
        t.nbytes = file_row_bytes(t.type, t.ne0) * (size_t)t.ne1;
        t.file_offset = (uint64_t)ftello(f);
        if (fseeko(f, (off_t)t.nbytes, SEEK_CUR) != 0)
            BG_FAIL(false, "invalid model file '%s' (seek failed)", fname.c_str());
        by_name[t.name] = (int)tensors.size();
        tensors.push_back(std::move(t));
    }
    // the last payload must be fully present
    if (!tensors.empty()) {
        fseeko(f, 0, SEEK_END);
        const uint64_t size = (uint64_t)ftello(f);
        const TensorEntry &last = tensors.back();
        if (last.file_offset + last.nbytes > size)
            BG_FAIL(false, "tensor '%s' has wrong size in model file (truncated payload)", last.name.c_str());
    }
    return true;
}

bool ModelFile::read_payload(const TensorEntry &t, void *dst) const {
    File fh;
    fh.f = fopen(path.c_str(), "rb");
    if (!fh.f) BG_FAIL(false, "failed to open '%s'", path.c_str());
    if (fseeko(fh.f, (off_t)t.file_offset, SEEK_SET) != 0) BG_FAIL(false, "seek failed in '%s'", path.c_str());
    if (fread(dst, 1, t.nbytes, fh.f) != t.nbytes)
        BG_FAIL(false, "tensor '%s' truncated in '%s'", t.name.c_str(), path.c_str());
    return true;
}

}  // namespace bg
