"""Pins the oracle against every known-answer value the reference's own tests hold for this path
 and against published vectors of the un-vendored third-party algorithms."""
import hashlib

import mpvss_oracle as O


# ---- reference KATs -------------------------------------------------------------------------

def test_polynomial_get_value_kats():
    # polynomial.rs:75-108
    c = [3, 2, 2, 4]
    assert [O.poly_get_value(c, x) for x in (0, 1, 2, 3)] == [3, 11, 47, 135]
    # polynomial.rs:111-125
    c = [105211, 1548877, 892134, 3490857, 324, 14234735]
    assert O.poly_get_value(c, 278) % 15486967 == 4115179


def test_util_kats():
    # util.rs:84-115
    assert O.extend_gcd(26, 3) == (1, -1, 9)
    assert O.mod_inverse(3, 26) == 9
    assert O.mod_inverse(4, 32) is None
    # util.rs:118-138
    values = [0, 1, 2, 3, 4, 5, 6]
    assert O.lagrange_coefficient(9, values) == (0, 1)
    assert O.lagrange_coefficient(1, values) == (720, 120)
    assert O.lagrange_coefficient(2, values) == (360, -24)
    assert O.lagrange_coefficient(3, values) == (240, 12)
    assert O.lagrange_coefficient(3, [1, 3, 4]) == (4, -2)
    # util.rs:157-161
    assert 1337 ^ 42 == 1299


def test_sha256_kat_from_util_rs():
    # util.rs:164-190
    h = hashlib.sha256()
    h.update(b"43589072349864890574839")
    h.update(b"14735247304952934566")
    assert h.hexdigest() == "e25e5b7edf4ea66e5238393fb4f183e0fc1593c69a522f9255a51bd0bc2b7ba7"
    assert int(h.hexdigest(), 16) == \
        102389418883295205726805934198606438410316463205994911160958467170744727731111


def test_dleq_response_kat():
    # dleq.rs:380-403
    g = O.ModpGroup()
    r = O.dleq_response(g, 81647, 163027, 127997)
    assert r == (81647 - 163027 * 127997) % g.order()
    # dleq.rs:359-377: a1 = g1^w, a2 = g2^w
    assert g.exp(8443, 81647) == pow(8443, 81647, g.q)
    assert g.exp(1299721, 81647) == pow(1299721, 81647, g.q)


def test_modp_group_kats():
    g = O.ModpGroup()
    assert g.exp(g.generator(), 1) == 2 and g.exp(g.generator(), 0) == 1     # modp.rs:243-250
    assert g.mul(5, 3) == 15                                                     # modp.rs:253-259
    assert g.hash_to_scalar(b"test data") < g.subgroup_order()                  # modp.rs:262-268
    assert g.subgroup_generator() == 4 and g.order() == g.q - 1 and g.subgroup_order() == (g.q - 1) // 2
    assert g.element_to_bytes(0) == b"\x00" and g.element_to_bytes(1 << 36) == bytes([0x10, 0, 0, 0, 0])


def test_secp256k1_kats():
    s = O.Secp256k1Group()
    G = s.generator()
    # secp256k1.rs:197-205 order constant; 217-235 basic ops; 270-272 encoding length
    assert s.group_order_int() == 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    assert s.exp(G, 1) == G and s.exp(G, 0) is None
    assert s.mul(G, G) == s.exp(G, 2)
    assert len(s.element_to_bytes(G)) == 33
    # SEC2 / widely published multiples of G (also SURVEY appendix B via OpenSSL)
    assert s.element_to_bytes(s.exp(G, 2)).hex() == "02c6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5"
    assert s.element_to_bytes(s.exp(G, 3)).hex() == "02f9308a019258c31049344f85f89d5229b531c845836f99b08601f113bce036f9"
    assert s.element_to_bytes(s.exp(s.exp(G, 2), 3)).hex() == \
        "03fff97bd5755eeea420453a14355235d382f6472f8568a18b2f057a1460297556"
    assert s.exp(G, s.n) is None and s.exp(G, s.n - 1) == s.element_inverse(G)
    for k in (1, 2, 3, 77, s.n - 5):
        ok, P = s.decode_element(s.element_to_bytes(s.exp(G, k)))
        assert ok and P == s.exp(G, k)
    assert s.bytes_to_element(b"\x05" + bytes(32)) is None and s.bytes_to_element(bytes(32)) is None


def test_ristretto255_kats():
    r = O.Ristretto255Group()
    B = r.generator()
    # ristretto255.rs:378-401 order constant
    assert r.group_order_int() == 2**252 + 27742317777372353535851937790883648493
    # RFC 9496 appendix A.1: multiples of the generator
    expected = [
        "0000000000000000000000000000000000000000000000000000000000000000",
        "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
        "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
        "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
        "da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57",
        "e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e",
        "f64746d3c92b13050ed8d80236a7f0007c3b3f962f5ba793d19a601ebb1df403",
        "44f53520926ec81fbd5a387845beb7df85a96a24ece18738bdcfa6a7822a176d",
        "903293d8f2287ebe10e2374dc1a53e0bc887e592699f02d077d5263cdd55601c",
        "02622ace8f7303a31cafc63f8fc48fdc16e1c8c8d234b2f0d6685282a9076031",
        "20706fd788b2720a1ed2a5dad4952b01f413bcf0e7564de8cdc816689e2db95f",
        "bce83f8ba5dd2fa572864c24ba1810f9522bc6004afe95877ac73241cafdab42",
        "e4549ee16b9aa03099ca208c67adafcafa4c3f3e4e5303de6026e3ca8ff84460",
        "aa52e000df2e16f55fb1032fc33bc42742dad6bd5a8fc0be0167436c5948501f",
        "46376b80f409b29dc2b5f6f0c52591990896e5716f41477cd30085ab7f10301e",
        "e0c418f7c8d9c4cdd7395b93ea124f3ad99021bb681dfc3302a9d99a2e53e64e",
    ]
    for k, e in enumerate(expected):
        P = r.exp(B, k)
        assert r.element_to_bytes(P).hex() == e
        Q = r.bytes_to_element(bytes.fromhex(e))
        assert Q is not None and r.elements_equal(P, Q)
    # RFC 9496 appendix A.2: a few invalid encodings
    for bad in ("00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",   # non-canonical field element
                "0100000000000000000000000000000000000000000000000000000000000000",   # negative field element
                "26948d35ca62e643e26a83177332e6b6afeb9d08e4268b650f1f5bbd8d81d371"):  # non-square x^2
        assert r.bytes_to_element(bytes.fromhex(bad)) is None
    # ristretto255.rs:620-638 basic ops, 680-682 encoding length
    assert r.elements_equal(r.mul(B, B), r.exp(B, 2))
    assert r.element_to_bytes(r.exp(B, 0)) == bytes(32)
    assert len(r.element_to_bytes(B)) == 32
    # bigint <-> scalar conversion semantics (ristretto255.rs:78-125)
    assert r.scalar_from_bigint(5) == 5 and r.scalar_from_bigint(r.l + 7) == 7
    assert r.scalar_to_bytes(1) == b"\x01" + bytes(31)


# ---- SURVEY appendix B (computed independently of oracle/mpvss_oracle.py) ----------------------

def test_survey_appendix_b_modp_distribution():
    g = O.ModpGroup()
    pks = [g.generate_public_key(k) for k in (5, 9)]
    box = O.distribute_secret(g, 0x4869, pks, 2, [7, 11], [13, 17])
    assert box["commitments"] == [pow(4, 7, g.q), pow(4, 11, g.q)]
    assert box["_X"][0] == 1 << 36 and box["shares"][g.element_to_bytes(pks[0])] == 1 << 90
    assert box["_digest"].hex() == "5b1da312869d00956a82fb60dc26147ec3d37e2b751fa633f11f74b8f567e333"
    assert box["challenge"] == 0x16bf4fb45e555856a2c043e449380306ee499748d9369495db90a2f429c802ab
    r = [box["responses"][g.element_to_bytes(p)] for p in pks]
    assert hashlib.sha256(r[0].to_bytes(256, "big")).hexdigest() == \
        "c05cc33c5b91c2bf97a148279ed273807e98070e0a0f62cdb5c087b21d4b9c4d"
    assert hashlib.sha256(r[1].to_bytes(256, "big")).hexdigest() == \
        "39d1183def16e4a304cb967fd79661b8678ef2848569c366c03535c31eb8e037"
    assert box["U"] == 0x76be8b528d0075f7aae98d6fa57a6d3c83ae480a8469e668d7b0af968995e418
    assert O.verify_distribution_shares(g, box)


def test_horner_in_the_exponent_equals_reference_loop():
    """The kernels evaluate X_i by Horner's rule in the exponent; it must be the same element as the
    reference's loop (participant.rs:423-434) for units, zero and unreduced commitments."""
    import random
    g = O.ModpGroup()
    rng = random.Random(3)
    for commitments in ([rng.randrange(1 << 2048) for _ in range(4)], [0, 5, 9], [7, 0, 1], [g.q + 2, 3]):
        for i in (0, 1, 2, 5, 1000, 65536):
            acc = commitments[-1] % g.q
            for c in reversed(commitments[:-1]):
                acc = pow(acc, i, g.q) * c % g.q
            assert acc == O.commitment_eval(g, commitments, i)
