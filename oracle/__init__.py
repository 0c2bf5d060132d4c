"""CPU oracle for the NeuTTS synthesis hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
(`neutts-air_amd/`).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
reported baseline -- never as the thing measured or shipped.

Model geometries and the seeded random weights / prompts of the benchmark are plain data and live in `synthetic.py`
(re-exported by backbone_ref / codec_ref for the tests); `bench.py` and `tools/` take them from there and touch this
package only in the `cpu_baseline` leg.

Pinning status (see DESIGN.md section "Oracle"):
  * backbone_ref.py  -- pinned: bit-checked against transformers' Qwen2ForCausalLM
    (the un-vendored dependency the reference calls, ref:neutts/neutts.py:164,338-347)
    by tests/test_oracle_pin.py and against the committed fixtures in tests/golden/
    that oracle/gen_golden.py produced from that dependency.
  * codec_ref.py     -- pinned against transformers.models.xcodec2 at NeuCodec geometry
    (hop 480); the equivalence NeuCodec-decoder == xcodec2-decoder@hop480 itself cannot
    be verified offline (neucodec is not installable here): "parity unpinned" at that
    one boundary.
The reference's own tests hold no golden vectors for this path
(ref:tests/test_neutts.py:55-58 checks type/len/NaN/dtype only).
"""
