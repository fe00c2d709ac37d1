/* Kaiser window samples used by the resampler at quality 4 (the only quality the IAMF decoder
 * uses: SPEEX_RESAMPLER_QUALITY 4, reference src/iamf_dec/IAMF_decoder.c:57).  Coefficient DATA
 * of the speexdsp resampler (reference src/iamf_dec/resample.c:137-143: 36 samples of a Kaiser
 * window with beta ~ 8, read with 32x oversampling, stored as double there too). */
#define IAMF_RS_Q4_BASE_LENGTH 64
#define IAMF_RS_Q4_OVERSAMPLE 8
#define IAMF_RS_Q4_DOWN_BW 0.921f
#define IAMF_RS_Q4_UP_BW 0.940f
#define IAMF_RS_Q4_WINDOW_OVERSAMPLE 32
static const double iamf_rs_q4_window[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200,
    0.91223751, 0.87580811, 0.83439927, 0.78875245, 0.73966538, 0.68797126,
    0.63451750, 0.58014482, 0.52566725, 0.47185369, 0.41941150, 0.36897272,
    0.32108304, 0.27619388, 0.23465776, 0.19672670, 0.16255380, 0.13219758,
    0.10562887, 0.08273982, 0.06335451, 0.04724088, 0.03412321, 0.02369490,
    0.01563093, 0.00959968, 0.00527363, 0.00233883, 0.00050000, 0.00000000};
