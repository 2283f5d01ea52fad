// Swift side of the libq3asr C ABI (include/q3asr.h): what Qwen3ASRModel (Sources/Qwen3ASR/Qwen3ASR.swift) calls instead
// of its MLX modules.  fromPretrained (:693-770) keeps its download / cache / config.json parsing and its safetensors
// reader, constructs a Q3Backend, hands over every tensor by its checkpoint key and calls finalize(); transcribe
// (:60-145) keeps its signature and calls transcribeTokens for the part _generateText (:170-632) did.
//
// Never compiled in this repository (no Swift toolchain in its build image).
import CQ3ASR
import Foundation

public struct Q3Error: Error, CustomStringConvertible {
    public let message: String
    public var description: String { message }
    static func last() -> Q3Error { Q3Error(message: String(cString: q3asr_last_error())) }
}

public final class Q3Backend {
    private var handle: OpaquePointer?

    /// Hyper-parameters come from TextDecoderConfig (Configuration.swift:46-110) and Qwen3AudioEncoderConfig
    /// (AudioEncoder.swift:7-43); `maxBatch` bounds how many utterances one call decodes together.
    public init(vocab: Int, hidden: Int, layers: Int, heads: Int, kvHeads: Int, headDim: Int, intermediate: Int,
                rmsEps: Float, ropeTheta: Float, ropeInterleaved: Bool, rotaryDim: Int, groupSize: Int, bits: Int,
                encDModel: Int, encHeads: Int, encFFN: Int, encLayers: Int, melBins: Int, encOutDim: Int, convChannels: Int,
                nWindow: Int, nWindowInfer: Int, layerNormEps: Float, quantized: Bool, maxBatch: Int = 1,
                device: Int32 = 0) throws {
        var c = q3asr_config()
        c.enc_d_model = Int32(encDModel); c.enc_heads = Int32(encHeads); c.enc_ffn = Int32(encFFN)
        c.enc_layers = Int32(encLayers); c.n_mels = Int32(melBins); c.enc_out_dim = Int32(encOutDim)
        c.conv_channels = Int32(convChannels); c.n_window = Int32(nWindow); c.n_window_infer = Int32(nWindowInfer)
        c.ln_eps = layerNormEps
        c.vocab = Int32(vocab); c.hidden = Int32(hidden); c.layers = Int32(layers); c.heads = Int32(heads)
        c.kv_heads = Int32(kvHeads); c.head_dim = Int32(headDim); c.inter = Int32(intermediate)
        c.rms_eps = rmsEps; c.rope_theta = ropeTheta
        c.rope_interleaved = ropeInterleaved ? 1 : 0; c.rotary_dim = Int32(rotaryDim)
        c.group_size = Int32(groupSize); c.bits = Int32(bits)
        c.weight_format = quantized ? 1 : 0
        // Qwen3ASR.swift:190-196: the rescale of the audio embeddings is on unless this variable says otherwise
        c.embed_scale = ProcessInfo.processInfo.environment["QWEN3_ASR_DISABLE_EMBED_SCALE"] == "1" ? 0 : 1
        c.max_batch = Int32(maxBatch)
        guard q3asr_create(&c, device, &handle) == 0 else { throw Q3Error.last() }
    }

    deinit { q3asr_destroy(handle) }

    /// One safetensors entry: `dtype` is Q3ASR_F32 / Q3ASR_BF16 / Q3ASR_U32, `data` its raw payload (borrowed for the call).
    public func load(name: String, dtype: Int32, shape: [Int64], data: UnsafeRawPointer) throws {
        try shape.withUnsafeBufferPointer { s in
            guard q3asr_load_tensor(handle, name, dtype, s.baseAddress, Int32(shape.count), data) == 0 else { throw Q3Error.last() }
        }
    }

    /// Throws when a key is missing or mis-shaped (the reference skips such keys silently, WeightLoading.swift:190-203).
    public func finalize() throws { guard q3asr_finalize(handle) == 0 else { throw Q3Error.last() } }

    /// WhisperFeatureExtractor.process (AudioPreprocessing.swift:330-344): [128][frames] log-mel of one clip.
    public func logMel(_ audio: [Float], sampleRate: Int) throws -> (mel: [Float], frames: Int) {
        var offsets: [Int64] = [0, Int64(audio.count)]
        var out = [Float](repeating: 0, count: 128 * 3000)
        var frames: Int32 = 0
        let rc = audio.withUnsafeBufferPointer { a in
            q3asr_mel(handle, a.baseAddress, &offsets, 1, Int32(sampleRate), Int32(Q3ASR_HOST), &out, 3000, &frames,
                      Int32(Q3ASR_HOST))
        }
        guard rc == 0 else { throw Q3Error.last() }
        return (out, Int(frames))
    }

    /// One or many utterances; language prefixes (" language X<asr_text>", Qwen3ASR.swift:302-309) are tokenised by the
    /// caller's Qwen3Tokenizer as today.  Returns the generated ids of every utterance.
    public func transcribeTokens(_ clips: [[Float]], sampleRate: Int, languagePrefixes: [[Int32]]? = nil,
                                 maxTokens: Int = 448) throws -> [[Int32]] {
        var offsets: [Int64] = [0]
        for c in clips { offsets.append(offsets.last! + Int64(c.count)) }
        let audio = clips.flatMap { $0 }
        var tokens = [Int32](repeating: 0, count: clips.count * maxTokens)
        var counts = [Int32](repeating: 0, count: clips.count)
        var langIds: [Int32] = [], langOff: [Int32] = [0]
        for p in languagePrefixes ?? [] { langIds += p; langOff.append(Int32(langIds.count)) }
        let rc: Int32 = audio.withUnsafeBufferPointer { a in offsets.withUnsafeBufferPointer { o in
            tokens.withUnsafeMutableBufferPointer { t in counts.withUnsafeMutableBufferPointer { n in
            langIds.withUnsafeBufferPointer { li in langOff.withUnsafeBufferPointer { lo in
                var req = q3asr_request()
                req.audio = a.baseAddress; req.offsets = o.baseAddress; req.n_utt = Int32(clips.count)
                req.sample_rate = Int32(sampleRate); req.audio_space = Int32(Q3ASR_HOST)
                if languagePrefixes != nil { req.lang_ids = li.baseAddress; req.lang_offsets = lo.baseAddress }
                req.max_tokens = Int32(maxTokens); req.stop_rules = 1
                var res = q3asr_result()
                res.tokens = t.baseAddress; res.n_tokens = n.baseAddress
                return q3asr_transcribe_tokens(handle, &req, &res)
            }}}}}}
        guard rc == 0 else { throw Q3Error.last() }
        return (0..<clips.count).map { Array(tokens[$0 * maxTokens ..< $0 * maxTokens + Int(counts[$0])]) }
    }
}

// In Qwen3ASRModel (Qwen3ASR.swift:60-145), transcribe keeps its signature and stays non-throwing (:132-135):
//
//   public func transcribe(audio: [Float], sampleRate: Int = 24000, language: String? = nil, maxTokens: Int = 448) -> String {
//       let prefix = language.flatMap { lang in tokenizer.map { [$0.encode(" language \(lang)<asr_text>").map(Int32.init)] } }
//       guard let ids = try? backend.transcribeTokens([audio], sampleRate: sampleRate, languagePrefixes: prefix,
//                                                     maxTokens: maxTokens).first else { return "" }
//       guard let tokenizer else { return ids.map(String.init).joined(separator: " ") }      // :647-650
//       return postProcessASROutput(tokenizer.decode(tokens: ids.map(Int.init)))            // :639-662
//   }
