/*
 * TEST INFRASTRUCTURE ONLY (checker): CPU restatement of the reference's
 * Canvas2D colour conversion, CanvasRenderer.prototype.YCbCrToRGBA
 * (reference src/canvas2d.js:53-122), for the renderer-stage parity tests of
 * k_rgba.  Never linked into or called by the product.  Pinned against the
 * reference itself (oracle/ref_node_rgba.js runs the unmodified canvas2d.js
 * under Node; tests/golden/rgba_*.json hold the agreed md5s).
 *
 * The reference's render(y, cb, cr) is CALLED with (Y, Cr, Cb)
 * (src/mpeg1.js:235, src/mpeg1-wasm.js:118): its parameter `cb` receives the
 * Cr plane.  This function takes the planes by their true names.
 */
#include <stdint.h>
#include <string.h>

static uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* rgba: width * height * 4 bytes.  Like the reference after resize() (canvas2d.js:26-34: the buffer is
 * filled with 255) followed by YCbCrToRGBA: alpha stays 255, and so does everything in an odd last
 * column / row, which the 2x2 loop never reaches. */
void ycbcr_oracle_to_rgba(const uint8_t *y, const uint8_t *cr, const uint8_t *cb, int width, int height, uint8_t *rgba) {
	memset(rgba, 255, (size_t)width * height * 4);                 /* JSMpeg.Fill(imageData.data, 255) */
	const int w = ((width + 15) >> 4) << 4, w2 = w >> 1;           /* canvas2d.js:64-65 */
	int y_index1 = 0, y_index2 = w, y_next2 = w + (w - width);     /* :67-69 */
	int c_index = 0, c_next = w2 - (width >> 1);                   /* :71-72 */
	int o1 = 0, o2 = width * 4, o_next2 = width * 4;               /* :74-76 */
	const int cols = width >> 1, rows = height >> 1;               /* :78-79 */
	for (int row = 0; row < rows; row++) {
		for (int col = 0; col < cols; col++) {
			const int ccb = cr[c_index], ccr = cb[c_index];         /* the reference's names: ccb <- its 2nd argument */
			c_index++;
			const int r = (ccb + ((ccb * 103) >> 8)) - 179;         /* :88-90 */
			const int g = ((ccr * 88) >> 8) - 44 + ((ccb * 183) >> 8) - 91;
			const int b = (ccr + ((ccr * 198) >> 8)) - 227;
			const int y1 = y[y_index1++], y2 = y[y_index1++];       /* :93-101 */
			rgba[o1] = clamp_u8(y1 + r); rgba[o1 + 1] = clamp_u8(y1 - g); rgba[o1 + 2] = clamp_u8(y1 + b);
			rgba[o1 + 4] = clamp_u8(y2 + r); rgba[o1 + 5] = clamp_u8(y2 - g); rgba[o1 + 6] = clamp_u8(y2 + b);
			o1 += 8;
			const int y3 = y[y_index2++], y4 = y[y_index2++];       /* :104-112 */
			rgba[o2] = clamp_u8(y3 + r); rgba[o2 + 1] = clamp_u8(y3 - g); rgba[o2 + 2] = clamp_u8(y3 + b);
			rgba[o2 + 4] = clamp_u8(y4 + r); rgba[o2 + 5] = clamp_u8(y4 - g); rgba[o2 + 6] = clamp_u8(y4 + b);
			o2 += 8;
		}
		y_index1 += y_next2; y_index2 += y_next2;                   /* :115-119 */
		o1 += o_next2; o2 += o_next2;
		c_index += c_next;
	}
}

/* The reference's WebGL renderer (src/webgl.js), restated in float64: textures of coded width x display height (luma) and
 * half that (chroma, height >> 1 rows) with GL_LINEAR / CLAMP_TO_EDGE (webgl.js:126-141, 203-215), viewport coded width x
 * height (webgl.js:122-124), texCoord = vertex (webgl.js:293-301), fragment shader webgl.js:259-281:
 * gl_FragColor = vec4(y, cr, cb, 1) * rec601 with the shader's `cb` holding the true Cr plane (render(y, cb, cr) is
 * called with (Y, Cr, Cb)).  PARITY UNPINNED for this form: no WebGL implementation runs in the build container, and a
 * browser's output depends on its GPU (shader precision `mediump`, filter precision); the GPU test compares the device
 * kernel with this restatement within 1 LSB. */
#include <math.h>
void ycbcr_oracle_to_rgba_gl(const uint8_t *y, const uint8_t *cr, const uint8_t *cb, int width, int height, uint8_t *rgba) {
	const int cw = ((width + 15) >> 4) << 4, cw2 = cw >> 1, h2 = height >> 1;
	for (int py = 0; py < height; py++)
		for (int px = 0; px < width; px++) {
			const double yv = y[(size_t)py * cw + px] / 255.0;
			double c_r = 0.5, c_b = 0.5;
			if (h2 > 0) {
				const double u = (px + 0.5) / cw * cw2 - 0.5, v = (py + 0.5) / height * h2 - 0.5;
				const double fu = floor(u), fv = floor(v), ax = u - fu, ay = v - fv;
				int x0 = (int)fu, x1 = x0 + 1, y0 = (int)fv, y1 = y0 + 1;
				if (x0 < 0) x0 = 0;
				if (x1 > cw2 - 1) x1 = cw2 - 1;
				if (y0 < 0) y0 = 0;
				if (y1 > h2 - 1) y1 = h2 - 1;
#define SAMPLE(P) ((1.0 - ay) * ((1.0 - ax) * P[(size_t)y0 * cw2 + x0] + ax * P[(size_t)y0 * cw2 + x1]) + ay * ((1.0 - ax) * P[(size_t)y1 * cw2 + x0] + ax * P[(size_t)y1 * cw2 + x1])) / 255.0
				c_r = SAMPLE(cr); c_b = SAMPLE(cb);
#undef SAMPLE
			}
			double R = 1.16438 * yv + 1.59603 * c_r - 0.87079;
			double G = 1.16438 * yv - 0.39176 * c_b - 0.81297 * c_r + 0.52959;
			double B = 1.16438 * yv + 2.01723 * c_b - 1.08139;
			R = R < 0 ? 0 : (R > 1 ? 1 : R); G = G < 0 ? 0 : (G > 1 ? 1 : G); B = B < 0 ? 0 : (B > 1 ? 1 : B);
			uint8_t *o = rgba + ((size_t)py * width + px) * 4;
			o[0] = (uint8_t)(R * 255.0 + 0.5); o[1] = (uint8_t)(G * 255.0 + 0.5); o[2] = (uint8_t)(B * 255.0 + 0.5); o[3] = 255;
		}
}
