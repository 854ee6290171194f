// Bundle writer (host side): finished games -> one `bundle_<uuid>.npz`, the on-disk format the
// reference's training pipeline reads (alpharat/data/loader.py:114-231).
//   crates/alpharat-sampling/src/npz_writer.rs:54-176  .npy v1.0, 256-byte header, deflate zip
//   crates/alpharat-sampling/src/recording.rs:23-162   26 arrays, names / dtypes / shapes / order
//   crates/alpharat-sampling/src/recording.rs:121,161  atomic tmp -> rename
#pragma once
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#include "../../include/alpharat_hip.h"

namespace ar {

class NpzWriter {
  public:
    bool open(const std::string& path, std::string& err) {
        f_ = fopen(path.c_str(), "wb");
        if (!f_) {
            err = "cannot create " + path + ": " + strerror(errno);
            return false;
        }
        return true;
    }
    // descr e.g. "<f4", "|i1", "<i2", "<i4", "|b1"
    bool add(const char* name, const char* descr, const std::vector<size_t>& shape, const void* data, size_t bytes,
             std::string& err) {
        unsigned char header[256];
        memset(header, ' ', sizeof header);
        std::string dict = std::string("{'descr':'") + descr + "','fortran_order':False,'shape':" + shape_str(shape) + "}";
        if (dict.size() >= 246) {
            err = "npy header too long";
            return false;
        }
        header[0] = 0x93;
        memcpy(header + 1, "NUMPY", 5);
        header[6] = 1;
        header[7] = 0;
        header[8] = 246;  // 256 - 10, little endian u16
        header[9] = 0;
        memcpy(header + 10, dict.data(), dict.size());
        header[255] = '\n';
        std::vector<unsigned char> raw(256 + bytes);
        memcpy(raw.data(), header, 256);
        if (bytes) memcpy(raw.data() + 256, data, bytes);
        // raw deflate
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
            err = "deflateInit2 failed";
            return false;
        }
        std::vector<unsigned char> comp(deflateBound(&zs, (uLong)raw.size()));
        zs.next_in = raw.data();
        zs.avail_in = (uInt)raw.size();
        zs.next_out = comp.data();
        zs.avail_out = (uInt)comp.size();
        int rc = deflate(&zs, Z_FINISH);
        size_t csize = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) {
            err = "deflate failed";
            return false;
        }
        Entry e;
        e.name = std::string(name) + ".npy";
        e.crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
        e.csize = (uint32_t)csize;
        e.usize = (uint32_t)raw.size();
        e.offset = (uint32_t)ftell(f_);
        // local file header
        put32(0x04034b50);
        put16(20);
        put16(0);
        put16(8);
        put16(0);
        put16(0x21);  // time, date (1980-01-01)
        put32(e.crc);
        put32(e.csize);
        put32(e.usize);
        put16((uint16_t)e.name.size());
        put16(0);
        fwrite(e.name.data(), 1, e.name.size(), f_);
        fwrite(comp.data(), 1, csize, f_);
        entries_.push_back(e);
        if (ferror(f_)) {
            err = "write error";
            return false;
        }
        return true;
    }
    bool finish(std::string& err) {
        uint32_t cd_start = (uint32_t)ftell(f_);
        for (const Entry& e : entries_) {
            put32(0x02014b50);
            put16(20);
            put16(20);
            put16(0);
            put16(8);
            put16(0);
            put16(0x21);
            put32(e.crc);
            put32(e.csize);
            put32(e.usize);
            put16((uint16_t)e.name.size());
            put16(0);
            put16(0);
            put16(0);
            put16(0);
            put32(0);
            put32(e.offset);
            fwrite(e.name.data(), 1, e.name.size(), f_);
        }
        uint32_t cd_size = (uint32_t)ftell(f_) - cd_start;
        put32(0x06054b50);
        put16(0);
        put16(0);
        put16((uint16_t)entries_.size());
        put16((uint16_t)entries_.size());
        put32(cd_size);
        put32(cd_start);
        put16(0);
        bool ok = !ferror(f_);
        ok = (fclose(f_) == 0) && ok;
        f_ = nullptr;
        if (!ok) err = "write error while finishing archive";
        return ok;
    }
    ~NpzWriter() {
        if (f_) fclose(f_);
    }

  private:
    struct Entry {
        std::string name;
        uint32_t crc, csize, usize, offset;
    };
    static std::string shape_str(const std::vector<size_t>& s) {
        if (s.empty()) return "()";
        if (s.size() == 1) return "(" + std::to_string(s[0]) + ",)";
        std::string r = "(";
        for (size_t i = 0; i < s.size(); ++i) r += (i ? "," : "") + std::to_string(s[i]);
        return r + ")";
    }
    void put16(uint16_t v) { fwrite(&v, 2, 1, f_); }
    void put32(uint32_t v) { fwrite(&v, 4, 1, f_); }
    FILE* f_ = nullptr;
    std::vector<Entry> entries_;
};

// recording.rs:23-162 write_bundle
inline bool write_bundle(const ArGameRecordView* games, uint32_t k, const std::string& path, std::string& err) {
    if (k == 0) {
        err = "no games to write";
        return false;
    }
    const size_t w = games[0].width, h = games[0].height, hw = w * h;
    size_t n = 0;
    for (uint32_t i = 0; i < k; ++i) {
        if (games[i].width != w || games[i].height != h) {
            err = "game " + std::to_string(i) + " has dimensions " + std::to_string(games[i].width) + "x" +
                  std::to_string(games[i].height) + ", expected " + std::to_string(w) + "x" + std::to_string(h);
            return false;
        }
        if (games[i].n_positions == 0) {
            err = "game " + std::to_string(i) + " has no positions";
            return false;
        }
        n += games[i].n_positions;
    }
    std::vector<int32_t> game_lengths(k);
    std::vector<int8_t> maze(k * hw * 4), cheese_outcomes(k * hw), result(k);
    std::vector<uint8_t> initial_cheese(k * hw), cheese_mask(n * hw);
    std::vector<int16_t> max_turns(k), turn(n);
    std::vector<float> fp1(k), fp2(k), p1_score(n), p2_score(n), value_p1(n), value_p2(n);
    std::vector<float> vc1(n * 5), vc2(n * 5), pr1(n * 5), pr2(n * 5), po1(n * 5), po2(n * 5);
    std::vector<int8_t> p1_pos(n * 2), p2_pos(n * 2), p1_mud(n), p2_mud(n), action_p1(n), action_p2(n);
    size_t at = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const ArGameRecordView& g = games[i];
        const size_t m = g.n_positions;
        game_lengths[i] = (int32_t)m;
        memcpy(&maze[i * hw * 4], g.maze, hw * 4);
        memcpy(&initial_cheese[i * hw], g.initial_cheese, hw);
        for (size_t c = 0; c < hw; ++c) cheese_outcomes[i * hw + c] = (int8_t)g.cheese_outcomes[c];
        max_turns[i] = (int16_t)g.max_turns;
        result[i] = (int8_t)g.result;
        fp1[i] = g.final_p1_score;
        fp2[i] = g.final_p2_score;
        for (size_t j = 0; j < m; ++j, ++at) {
            p1_pos[at * 2] = (int8_t)g.p1_pos[j * 2];
            p1_pos[at * 2 + 1] = (int8_t)g.p1_pos[j * 2 + 1];
            p2_pos[at * 2] = (int8_t)g.p2_pos[j * 2];
            p2_pos[at * 2 + 1] = (int8_t)g.p2_pos[j * 2 + 1];
            p1_score[at] = g.p1_score[j];
            p2_score[at] = g.p2_score[j];
            p1_mud[at] = (int8_t)g.p1_mud[j];
            p2_mud[at] = (int8_t)g.p2_mud[j];
            memcpy(&cheese_mask[at * hw], g.cheese_mask + j * hw, hw);
            turn[at] = (int16_t)g.turn[j];
            value_p1[at] = g.value_p1[j];
            value_p2[at] = g.value_p2[j];
            memcpy(&vc1[at * 5], g.visit_counts_p1 + j * 5, 20);
            memcpy(&vc2[at * 5], g.visit_counts_p2 + j * 5, 20);
            memcpy(&pr1[at * 5], g.prior_p1 + j * 5, 20);
            memcpy(&pr2[at * 5], g.prior_p2 + j * 5, 20);
            memcpy(&po1[at * 5], g.policy_p1 + j * 5, 20);
            memcpy(&po2[at * 5], g.policy_p2 + j * 5, 20);
            action_p1[at] = (int8_t)g.action_p1[j];
            action_p2[at] = (int8_t)g.action_p2[j];
        }
    }
    const std::string tmp = path.substr(0, path.size() >= 4 && path.compare(path.size() - 4, 4, ".npz") == 0
                                               ? path.size() - 4
                                               : path.size()) +
                            ".npz.tmp";
    NpzWriter z;
    if (!z.open(tmp, err)) return false;
    bool ok = true;
#define ADD(name, descr, shape, vec) ok = ok && z.add(name, descr, shape, vec.data(), vec.size() * sizeof(vec[0]), err)
    typedef std::vector<size_t> S;
    ADD("game_lengths", "<i4", S({k}), game_lengths);
    ADD("maze", "|i1", S({k, h, w, 4}), maze);
    ADD("initial_cheese", "|b1", S({k, h, w}), initial_cheese);
    ADD("cheese_outcomes", "|i1", S({k, h, w}), cheese_outcomes);
    ADD("max_turns", "<i2", S({k}), max_turns);
    ADD("result", "|i1", S({k}), result);
    ADD("final_p1_score", "<f4", S({k}), fp1);
    ADD("final_p2_score", "<f4", S({k}), fp2);
    ADD("p1_pos", "|i1", S({n, 2}), p1_pos);
    ADD("p2_pos", "|i1", S({n, 2}), p2_pos);
    ADD("p1_score", "<f4", S({n}), p1_score);
    ADD("p2_score", "<f4", S({n}), p2_score);
    ADD("p1_mud", "|i1", S({n}), p1_mud);
    ADD("p2_mud", "|i1", S({n}), p2_mud);
    ADD("cheese_mask", "|b1", S({n, h, w}), cheese_mask);
    ADD("turn", "<i2", S({n}), turn);
    ADD("value_p1", "<f4", S({n}), value_p1);
    ADD("value_p2", "<f4", S({n}), value_p2);
    ADD("visit_counts_p1", "<f4", S({n, 5}), vc1);
    ADD("visit_counts_p2", "<f4", S({n, 5}), vc2);
    ADD("prior_p1", "<f4", S({n, 5}), pr1);
    ADD("prior_p2", "<f4", S({n, 5}), pr2);
    ADD("policy_p1", "<f4", S({n, 5}), po1);
    ADD("policy_p2", "<f4", S({n, 5}), po2);
    ADD("action_p1", "|i1", S({n}), action_p1);
    ADD("action_p2", "|i1", S({n}), action_p2);
#undef ADD
    ok = ok && z.finish(err);
    if (!ok) {
        remove(tmp.c_str());
        return false;
    }
    if (rename(tmp.c_str(), path.c_str()) != 0) {
        err = "rename failed: " + std::string(strerror(errno));
        remove(tmp.c_str());
        return false;
    }
    return true;
}

}  // namespace ar
