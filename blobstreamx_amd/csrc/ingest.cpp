// ingest.cpp — wire-format ingest: Tendermint RPC JSON -> the packed device layouts of bsx.h.
//
// Replaces, for the data path, what the reference does with serde + tendermint-rs when it reads
// `/commit`, `/validators`, `signed_block.json` and `data_commitment.json` (circuits/input.rs:19-27,67-110,
// 120-145; circuits/fetcher.rs:44-58,89-132; the fixture layout of circuits/fixtures/mocha-4/**):
//   header JSON        -> the 14 protobuf-encoded Merkle leaves of `Header::hash` (bsx_header)
//   commit + validators -> one bsx_validator per validator: pubkey, voting power, signature and the CanonicalVote
//                          sign-bytes each validator signed (SURVEY Appendix A byte formats)
// Pure host-side byte formatting (RFC 3339 -> protobuf Timestamp, varints, base64, hex); no hashing, no GPU needed.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bsx.h"

namespace {

// ------------------------------------------------------------------ minimal JSON DOM
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    std::string s;                                  // Str: decoded text; Num: literal text
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct Parser {
    const char* p; const char* e; bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    bool lit(const char* t) { size_t n = strlen(t); if ((size_t)(e - p) >= n && !memcmp(p, t, n)) { p += n; return true; } return false; }
    std::string str() {
        std::string out;
        if (p >= e || *p != '"') { ok = false; return out; }
        p++;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                p++;
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': { if (e - p < 5) { ok = false; return out; } unsigned v = 0; sscanf(std::string(p + 1, 4).c_str(), "%x", &v); if (v < 0x80) out += (char)v; else out += '?'; p += 4; break; }
                    default: out += *p;
                }
                p++;
            } else out += *p++;
        }
        if (p >= e) { ok = false; return out; }
        p++;
        return out;
    }
    JVal val(int depth = 0) {
        JVal v;
        if (depth > 64) { ok = false; return v; }
        ws();
        if (p >= e) { ok = false; return v; }
        if (*p == '{') {
            v.kind = JVal::Obj; p++; ws();
            if (p < e && *p == '}') { p++; return v; }
            while (ok) {
                ws(); std::string k = str(); ws();
                if (!ok || p >= e || *p != ':') { ok = false; break; }
                p++;
                v.obj.emplace_back(std::move(k), val(depth + 1));
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; break; }
                ok = false;
            }
        } else if (*p == '[') {
            v.kind = JVal::Arr; p++; ws();
            if (p < e && *p == ']') { p++; return v; }
            while (ok) {
                v.arr.push_back(val(depth + 1)); ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; break; }
                ok = false;
            }
        } else if (*p == '"') { v.kind = JVal::Str; v.s = str(); }
        else if (lit("true")) { v.kind = JVal::Bool; v.b = true; }
        else if (lit("false")) { v.kind = JVal::Bool; }
        else if (lit("null")) { v.kind = JVal::Null; }
        else {
            v.kind = JVal::Num;
            const char* s0 = p;
            while (p < e && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) p++;
            if (p == s0) ok = false;
            v.s.assign(s0, p);
        }
        return v;
    }
};

thread_local std::string g_ingest_err;
int ifail(int code, const std::string& m) { g_ingest_err = m; return code; }

// ------------------------------------------------------------------ small codecs
bool as_u64(const JVal* v, uint64_t& out) {          // Tendermint encodes 64-bit ints as strings, small ones as numbers
    if (!v || (v->kind != JVal::Str && v->kind != JVal::Num) || v->s.empty()) return false;
    uint64_t x = 0;
    for (char c : v->s) {
        if (c < '0' || c > '9') return false;
        const uint64_t d = (uint64_t)(c - '0');
        if (x > (0x7fffffffffffffffull - d) / 10) return false;      // Tendermint integers are int64: anything above is malformed
        x = x * 10 + d;
    }
    out = x;
    return true;
}
bool hex_decode(const std::string& s, std::vector<uint8_t>& out) {
    if (s.size() % 2) return false;
    out.clear();
    auto nib = [](char c) -> int { if (c >= '0' && c <= '9') return c - '0'; if (c >= 'a' && c <= 'f') return c - 'a' + 10; if (c >= 'A' && c <= 'F') return c - 'A' + 10; return -1; };
    for (size_t i = 0; i < s.size(); i += 2) { int a = nib(s[i]), b = nib(s[i + 1]); if (a < 0 || b < 0) return false; out.push_back((uint8_t)(a * 16 + b)); }
    return true;
}
bool b64_decode(const std::string& s, std::vector<uint8_t>& out) {
    out.clear();
    uint32_t acc = 0; int bits = 0;
    for (char c : s) {
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+') v = 62; else if (c == '/') v = 63; else if (c == '=') break; else return false;
        acc = acc << 6 | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}
void put_varint(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }

// RFC 3339 "YYYY-MM-DDTHH:MM:SS[.fraction]Z" -> (unix seconds, nanos)
bool parse_time(const std::string& s, int64_t& secs, uint32_t& nanos) {
    int Y, M, D, h, m, sec;
    if (s.size() < 20 || sscanf(s.c_str(), "%4d-%2d-%2dT%2d:%2d:%2d", &Y, &M, &D, &h, &m, &sec) != 6) return false;
    size_t i = 19;
    nanos = 0;
    if (i < s.size() && s[i] == '.') {
        i++;
        uint32_t f = 0; int nd = 0;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') { if (nd < 9) { f = f * 10 + (uint32_t)(s[i] - '0'); nd++; } i++; }
        while (nd < 9) { f *= 10; nd++; }
        nanos = f;
    }
    if (i >= s.size() || s[i] != 'Z') return false;      // Tendermint always emits UTC
    // days from civil (proleptic Gregorian)
    int y = Y - (M <= 2);
    const int era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153u * (unsigned)(M + (M > 2 ? -3 : 9)) + 2) / 5 + (unsigned)D - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    const int64_t days = (int64_t)era * 146097 + (int64_t)doe - 719468;
    secs = days * 86400 + h * 3600 + m * 60 + sec;
    return true;
}
std::vector<uint8_t> enc_timestamp(int64_t secs, uint32_t nanos) {
    std::vector<uint8_t> o;
    if (secs) { o.push_back(0x08); put_varint(o, (uint64_t)secs); }
    if (nanos) { o.push_back(0x10); put_varint(o, nanos); }
    return o;
}
std::vector<uint8_t> enc_bytes_value(const std::vector<uint8_t>& b) {
    std::vector<uint8_t> o;
    if (b.empty()) return o;
    o.push_back(0x0a); put_varint(o, b.size()); o.insert(o.end(), b.begin(), b.end());
    return o;
}
bool put_field(bsx_header* h, int idx, const std::vector<uint8_t>& f) {
    static const int cap[14] = {24, 52, 12, 20, 76, 36, 36, 36, 36, 36, 36, 36, 36, 24};
    if ((int)f.size() > cap[idx]) return false;
    uint8_t* dst = idx == 0 ? h->version : idx == 1 ? h->chain_id : idx == 2 ? h->height : idx == 3 ? h->time
                 : idx == 4 ? h->last_block_id : idx == 13 ? h->proposer : h->hash[idx - 5];
    memcpy(dst, f.data(), f.size());
    h->len[idx] = (uint8_t)f.size();
    return true;
}

struct BlockId { std::vector<uint8_t> hash, parts_hash; uint64_t parts_total = 0; };
bool parse_block_id(const JVal* v, BlockId& out) {
    if (!v) return false;
    const JVal* h = v->get("hash"); const JVal* parts = v->get("parts");
    if (!parts) parts = v->get("part_set_header");
    if (!h || h->kind != JVal::Str || !hex_decode(h->s, out.hash) || !parts) return false;
    const JVal* ph = parts->get("hash");
    if (!ph || ph->kind != JVal::Str || !hex_decode(ph->s, out.parts_hash)) return false;
    return as_u64(parts->get("total"), out.parts_total);
}
std::vector<uint8_t> enc_block_id(const BlockId& b) {
    std::vector<uint8_t> psh, o;
    if (b.parts_total) { psh.push_back(0x08); put_varint(psh, b.parts_total); }
    if (!b.parts_hash.empty()) { psh.push_back(0x12); put_varint(psh, b.parts_hash.size()); psh.insert(psh.end(), b.parts_hash.begin(), b.parts_hash.end()); }
    if (!b.hash.empty()) { o.push_back(0x0a); put_varint(o, b.hash.size()); o.insert(o.end(), b.hash.begin(), b.hash.end()); }
    o.push_back(0x12); put_varint(o, psh.size()); o.insert(o.end(), psh.begin(), psh.end());
    return o;
}

int header_from_json(const JVal* hj, bsx_header* out, std::string& chain_id, uint64_t& height) {
    if (!hj || hj->kind != JVal::Obj) return ifail(BSX_ERR_BAD_ARG, "header object missing");
    memset(out, 0, sizeof *out);
    const JVal* ver = hj->get("version");
    uint64_t vb = 0, va = 0;
    if (ver) { as_u64(ver->get("block"), vb); as_u64(ver->get("app"), va); }
    std::vector<uint8_t> f;
    if (vb) { f.push_back(0x08); put_varint(f, vb); }
    if (va) { f.push_back(0x10); put_varint(f, va); }
    bool ok = put_field(out, 0, f);
    const JVal* cid = hj->get("chain_id");
    if (!cid || cid->kind != JVal::Str) return ifail(BSX_ERR_BAD_HEADER, "chain_id missing");
    chain_id = cid->s;
    ok = ok && put_field(out, 1, enc_bytes_value(std::vector<uint8_t>(chain_id.begin(), chain_id.end())));
    if (!as_u64(hj->get("height"), height)) return ifail(BSX_ERR_BAD_HEADER, "height missing");
    f.clear(); f.push_back(0x08); put_varint(f, height);
    ok = ok && put_field(out, 2, f);
    const JVal* t = hj->get("time");
    int64_t secs; uint32_t nanos;
    if (!t || t->kind != JVal::Str || !parse_time(t->s, secs, nanos)) return ifail(BSX_ERR_BAD_HEADER, "bad header time");
    ok = ok && put_field(out, 3, enc_timestamp(secs, nanos));
    BlockId lbi;
    if (!parse_block_id(hj->get("last_block_id"), lbi)) return ifail(BSX_ERR_BAD_HEADER, "bad last_block_id");
    ok = ok && put_field(out, 4, enc_block_id(lbi));
    static const char* names[8] = {"last_commit_hash", "data_hash", "validators_hash", "next_validators_hash", "consensus_hash",
                                   "app_hash", "last_results_hash", "evidence_hash"};
    for (int i = 0; i < 8; i++) {
        const JVal* hv = hj->get(names[i]);
        std::vector<uint8_t> b;
        if (!hv || hv->kind != JVal::Str || !hex_decode(hv->s, b)) return ifail(BSX_ERR_BAD_HEADER, std::string("bad ") + names[i]);
        ok = ok && put_field(out, 5 + i, enc_bytes_value(b));
    }
    const JVal* pa = hj->get("proposer_address");
    std::vector<uint8_t> b;
    if (!pa || pa->kind != JVal::Str || !hex_decode(pa->s, b)) return ifail(BSX_ERR_BAD_HEADER, "bad proposer_address");
    ok = ok && put_field(out, 13, enc_bytes_value(b));
    if (!ok) return ifail(BSX_ERR_BAD_HEADER, "an encoded header field exceeds its bsx_header capacity");
    return BSX_OK;
}

// CanonicalVote sign-bytes: varint(len) ‖ 08 02 ‖ 11 height LE ‖ [19 round LE] ‖ 22 .. block id ‖ 2a .. ts ‖ 32 .. chain id
bool sign_bytes(const std::string& chain_id, uint64_t height, uint64_t round, const BlockId& bid, int64_t secs, uint32_t nanos,
                std::vector<uint8_t>& out) {
    std::vector<uint8_t> body = {0x08, 0x02, 0x11};
    for (int i = 0; i < 8; i++) body.push_back((uint8_t)(height >> (8 * i)));
    if (round) { body.push_back(0x19); for (int i = 0; i < 8; i++) body.push_back((uint8_t)(round >> (8 * i))); }
    std::vector<uint8_t> psh = {0x08};
    put_varint(psh, bid.parts_total);
    psh.push_back(0x12); put_varint(psh, bid.parts_hash.size()); psh.insert(psh.end(), bid.parts_hash.begin(), bid.parts_hash.end());
    std::vector<uint8_t> cb = {0x0a};
    put_varint(cb, bid.hash.size()); cb.insert(cb.end(), bid.hash.begin(), bid.hash.end());
    cb.push_back(0x12); put_varint(cb, psh.size()); cb.insert(cb.end(), psh.begin(), psh.end());
    body.push_back(0x22); put_varint(body, cb.size()); body.insert(body.end(), cb.begin(), cb.end());
    std::vector<uint8_t> ts = enc_timestamp(secs, nanos);
    body.push_back(0x2a); put_varint(body, ts.size()); body.insert(body.end(), ts.begin(), ts.end());
    body.push_back(0x32); put_varint(body, chain_id.size()); body.insert(body.end(), chain_id.begin(), chain_id.end());
    out.clear();
    put_varint(out, body.size());
    out.insert(out.end(), body.begin(), body.end());
    return out.size() <= BSX_VALIDATOR_MSG_MAX;
}

}  // namespace

extern "C" {

const char* bsx_ingest_last_error(void) { return g_ingest_err.c_str(); }

// header.json / the "header" member of a /commit or signed_block response -> bsx_header
int bsx_ingest_header_json(const char* json, size_t len, bsx_header* out_header, uint64_t* out_height) {
    if (!json || !out_header) return ifail(BSX_ERR_BAD_ARG, "null pointer");
    Parser ps{json, json + len};
    JVal root = ps.val();
    if (!ps.ok) return ifail(BSX_ERR_BAD_ARG, "malformed JSON");
    const JVal* r = root.get("result") ? root.get("result") : &root;
    const JVal* hj = r->get("header");
    if (!hj && r->get("signed_header")) hj = r->get("signed_header")->get("header");
    std::string chain;
    uint64_t height = 0;
    int rc = header_from_json(hj, out_header, chain, height);
    if (rc == BSX_OK && out_height) *out_height = height;
    return rc;
}

// signed_block.json (fixture layout: result.{header, commit, validator_set}) or a /commit response
// (result.signed_header.{header, commit}, then validators_json = the /validators response) -> packed header +
// one bsx_validator per validator-set entry (slot order = validator-set order, as tendermintx feeds the circuit).
int bsx_ingest_signed_block_json(const char* json, size_t len, const char* validators_json, size_t validators_len,
                                 bsx_header* out_header, uint8_t out_block_hash[32], bsx_validator* out_validators,
                                 uint32_t v_max, uint32_t* out_n_validators, uint64_t* out_height) {
    if (!json || !out_header) return ifail(BSX_ERR_BAD_ARG, "null pointer");
    Parser ps{json, json + len};
    JVal root = ps.val();
    if (!ps.ok) return ifail(BSX_ERR_BAD_ARG, "malformed JSON");
    const JVal* r = root.get("result") ? root.get("result") : &root;
    const JVal* sh = r->get("signed_header") ? r->get("signed_header") : r;
    std::string chain;
    uint64_t height = 0;
    int rc = header_from_json(sh->get("header"), out_header, chain, height);
    if (rc) return rc;
    if (out_height) *out_height = height;
    const JVal* commit = sh->get("commit");
    if (!commit) return ifail(BSX_ERR_BAD_ARG, "commit missing");
    BlockId bid;
    if (!parse_block_id(commit->get("block_id"), bid) || bid.hash.size() != 32) return ifail(BSX_ERR_BAD_ARG, "bad commit.block_id");
    if (out_block_hash) memcpy(out_block_hash, bid.hash.data(), 32);
    uint64_t cheight = 0, round = 0;
    if (!as_u64(commit->get("height"), cheight)) return ifail(BSX_ERR_BAD_ARG, "commit.height missing or malformed");
    if (cheight != height) return ifail(BSX_ERR_BAD_ARG, "commit.height does not equal header.height");
    if (commit->get("round") && !as_u64(commit->get("round"), round)) return ifail(BSX_ERR_BAD_ARG, "commit.round malformed");
    if (!out_validators) { if (out_n_validators) *out_n_validators = 0; return BSX_OK; }

    JVal vroot;
    const JVal* vs = r->get("validator_set");
    if (vs) vs = vs->get("validators");
    if (!vs && validators_json) {
        Parser vp{validators_json, validators_json + validators_len};
        vroot = vp.val();
        if (!vp.ok) return ifail(BSX_ERR_BAD_ARG, "malformed validators JSON");
        const JVal* vr = vroot.get("result") ? vroot.get("result") : &vroot;
        vs = vr->get("validators");
    }
    if (!vs || vs->kind != JVal::Arr) return ifail(BSX_ERR_BAD_ARG, "validator set missing");
    if (vs->arr.size() > v_max) return ifail(BSX_ERR_RANGE_TOO_LONG, "validator set larger than MAX_VALIDATOR_SET_SIZE");
    const JVal* sigs = commit->get("signatures");
    memset(out_validators, 0, sizeof(bsx_validator) * (size_t)v_max);
    for (size_t i = 0; i < vs->arr.size(); i++) {
        const JVal& v = vs->arr[i];
        bsx_validator* o = &out_validators[i];
        const JVal* pk = v.get("pub_key");
        std::vector<uint8_t> pkb;
        if (!pk || !pk->get("value") || !b64_decode(pk->get("value")->s, pkb) || pkb.size() != 32) return ifail(BSX_ERR_BAD_ARG, "bad validator pub_key");
        memcpy(o->pubkey, pkb.data(), 32);
        uint64_t power = 0;
        if (!as_u64(v.get("voting_power"), power)) return ifail(BSX_ERR_BAD_ARG, "bad voting_power");
        o->voting_power = power;
        o->enabled = 1;
        o->present_on_trusted = 1;
        const JVal* addr = v.get("address");
        if (!sigs || sigs->kind != JVal::Arr || !addr) continue;
        for (const JVal& s : sigs->arr) {
            const JVal* sa = s.get("validator_address");
            uint64_t flag = 0;
            if (!as_u64(s.get("block_id_flag"), flag) || flag < 1 || flag > 3) return ifail(BSX_ERR_BAD_ARG, "bad block_id_flag");
            if (!sa || sa->s != addr->s || flag != 2) continue;      // 2 = BlockIDFlagCommit: signed this block id (1 absent, 3 nil)
            std::vector<uint8_t> sig;
            const JVal* sv = s.get("signature");
            const JVal* ts = s.get("timestamp");
            int64_t secs; uint32_t nanos;
            if (!sv || !b64_decode(sv->s, sig) || sig.size() != 64) return ifail(BSX_ERR_BAD_ARG, "bad signature encoding");
            if (!ts || !parse_time(ts->s, secs, nanos)) return ifail(BSX_ERR_BAD_ARG, "bad signature timestamp");
            std::vector<uint8_t> msg;
            if (!sign_bytes(chain, cheight, round, bid, secs, nanos, msg)) return ifail(BSX_ERR_BAD_ARG, "sign-bytes longer than 124 bytes");
            memcpy(o->signature, sig.data(), 64);
            memcpy(o->message, msg.data(), msg.size());
            o->message_len = (uint32_t)msg.size();
            o->is_signed = 1;
            break;
        }
    }
    if (out_n_validators) *out_n_validators = (uint32_t)vs->arr.size();
    return BSX_OK;
}

// data_commitment.json ({"result": {"data_commitment": "<hex>"}}, circuits/input.rs:19-27,104-109) -> 32 bytes
int bsx_ingest_data_commitment_json(const char* json, size_t len, uint8_t out[32]) {
    if (!json || !out) return ifail(BSX_ERR_BAD_ARG, "null pointer");
    Parser ps{json, json + len};
    JVal root = ps.val();
    if (!ps.ok) return ifail(BSX_ERR_BAD_ARG, "malformed JSON");
    const JVal* r = root.get("result") ? root.get("result") : &root;
    const JVal* d = r->get("data_commitment");
    std::vector<uint8_t> b;
    if (!d || d->kind != JVal::Str || !hex_decode(d->s, b) || b.size() != 32) return ifail(BSX_ERR_BAD_ARG, "bad data_commitment");
    memcpy(out, b.data(), 32);
    return BSX_OK;
}

}  // extern "C"
