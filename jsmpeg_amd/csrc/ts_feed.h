/*
 * The reference's MPEG-TS demuxer (src/ts.js:25-210) as HOST code in front of a live stream's write(): its state between
 * write() calls and the feed that turns TS bytes into destination.write(pts, payload) calls.  Shared by the live video streams
 * (live.hip: jsmpeg_hip_live_write_ts, jsmpeg_hip_ts_demux_host) and the live audio streams (mp2_live.hip:
 * jsmpeg_hip_mp2_live_write_ts).  Not installed; nothing outside jsmpeg_amd/csrc includes it.
 */
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "ts_sync.h"

/* a live stream fed as MPEG-TS (jsmpeg_hip_live_write_ts): the reference demuxer's state between write() calls (ts.js:3-41) */
struct LiveTs {
	std::vector<uint8_t> left;                         /* leftoverBytes */
	std::vector<std::pair<uint16_t, uint8_t>> pids;    /* pidsToStreamIds */
	uint32_t cur_len, total_len;                       /* pesPacketInfo[stream id]: currentLength, totalLength, pts, buffers */
	double pts;
	std::vector<uint8_t> pes;
	std::vector<uint8_t> joined;                       /* scratch: leftover + the new bytes */
	uint64_t writes;                                   /* destination.write calls made so far */
};

/* The stream as MPEG-TS: the reference's demuxer in front of write() (src/ts.js:25-147), with its state between calls --
 * leftover bytes of a cut packet (ts.js:25-41), the PID -> stream id table, the PES being collected (currentLength, totalLength,
 * pts) -- kept per live stream.  Host code like the ingest stage's framing pre-pass (ts_sync.h, shared): it looks at packet
 * HEADERS and moves payload bytes; every completed PES goes to `on_pes(pts, bytes, n)` (ts.js:189-194 packetComplete ->
 * destination.write(pts, buffers)).  Where the packets lie -- sync bytes, resync after garbage, what a write leaves over --
 * is jm_ts_sync_runs' restatement of ts.js:43-50, 150-187. */
template <class F>
static void live_ts_feed(LiveTs &T, const uint8_t *buf, uint64_t len, uint32_t stream_id, F &&on_pes) {
	if (!T.left.empty()) {
		T.joined.assign(T.left.begin(), T.left.end());
		T.joined.insert(T.joined.end(), buf, buf + len);
		buf = T.joined.data(); len = T.joined.size();
	}
	std::vector<JmTsRun> runs;
	uint64_t rest = 0;
	jm_ts_sync_runs(buf, len, nullptr, 0, runs, &rest);
	auto complete = [&]() {                                       /* ts.js:189-194 */
		on_pes(T.pts, T.pes.data(), (uint32_t)T.pes.size());
		T.writes++;
		T.total_len = 0; T.cur_len = 0; T.pes.clear();
	};
	for (const JmTsRun &r : runs) {
		for (uint32_t k = 0; k < r.packets; k++) {
			const uint8_t *p = buf + r.src + 188ull * k;
			const bool start = (p[1] & 0x40) != 0;
			const uint16_t pid = (uint16_t)(((p[1] & 0x1f) << 8) | p[2]);
			const uint32_t af = (p[3] >> 4) & 3u;
			uint32_t sid = 0;
			for (const auto &e : T.pids) if (e.first == pid) sid = e.second;
			if (start && sid == stream_id && T.cur_len) complete();        /* a new payload of the stream: the frame before it is over (ts.js:65-73) */
			if (!(af & 1)) continue;
			uint32_t at = 4;
			if (af & 2) at = 5u + p[4];
			if (at >= 188) continue;                                        /* (a header that runs past its packet: outside what a muxer writes; nothing of it is payload) */
			if (start && at + 9 <= 188 && p[at] == 0 && p[at + 1] == 0 && p[at + 2] == 1) {
				sid = p[at + 3];
				bool known = false;
				for (auto &e : T.pids) if (e.first == pid) { e.second = (uint8_t)sid; known = true; }
				if (!known) T.pids.push_back({ pid, (uint8_t)sid });
				const uint32_t packet_length = ((uint32_t)p[at + 4] << 8) | p[at + 5], flags = p[at + 7] >> 6, header_length = p[at + 8];
				if (sid == stream_id) {
					double pts = 0;
					if ((flags & 2) && at + 14 <= 188) {                    /* the 33-bit PTS in its five bytes (ts.js:96-113) */
						const uint8_t *q = p + at + 9;
						const double p32_30 = (q[0] >> 1) & 7, p29_15 = (((uint32_t)q[1] << 8) | q[2]) >> 1, p14_0 = (((uint32_t)q[3] << 8) | q[4]) >> 1;
						pts = (p32_30 * 1073741824.0 + p29_15 * 32768.0 + p14_0) / 90000.0;
					}
					T.total_len = packet_length ? packet_length - header_length - 3 : 0;      /* packetStart (ts.js:189-193) */
					T.cur_len = 0; T.pts = pts;
				}
				at += 9 + header_length;
			}
			if (sid != stream_id) continue;
			if (at < 188) { T.pes.insert(T.pes.end(), p + at, p + 188); T.cur_len += 188 - at; }
			const bool full = T.total_len != 0 && T.cur_len >= T.total_len;
			const bool padded = !start && (af & 2);                                     /* the video frame end guess (ts.js:127-147) */
			if (full || padded) complete();
		}
	}
	T.left.assign(buf + rest, buf + len);
}

