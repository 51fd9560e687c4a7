//! Same shapes as the reference crate's gate API (`hom_nand/src/tfhe.rs:20-71`, `hom_nand/src/tlwe.rs:19-79`):
//! inputs are moved, the output is owned, failure = panic.  Stable Rust (min_const_generics) is enough here: the
//! nightly features the reference needs are for its TRGSW array types, which stay behind the ABI.
use rtfhe_sys as sys;
use std::ffi::CStr;
use std::os::raw::c_int;

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Binary { Zero = 0, One = 1 }

/// `Torus32 = Decimal<u32>` (utils/src/math.rs:539): value x / 2^32, wrapping arithmetic.
pub type Torus32 = u32;

/// TLWE ciphertext (b, a[N]) -- `TLWERep<N>{cipher, p_key}` (hom_nand/src/tlwe.rs:19-23).
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct TLWERep<const N: usize> { cipher: Torus32, p_key: [Torus32; N] }

impl<const N: usize> TLWERep<N> {
    pub fn new(cipher: Torus32, p_key: [Torus32; N]) -> Self { TLWERep { cipher, p_key } }
    pub fn trivial(text: Torus32) -> Self { TLWERep { cipher: text, p_key: [0; N] } }
    pub fn logic_true() -> Self { Self::trivial(0x2000_0000) }   // AsLogic, tlwe.rs:80-87: +1/8
    pub fn logic_false() -> Self { Self::trivial(0xE000_0000) }  // -1/8
    pub fn cipher(&self) -> &Torus32 { &self.cipher }
    pub fn p_key(&self) -> &[Torus32; N] { &self.p_key }
    /// flat ABI layout: a[0..N) then b
    fn write_flat(&self, out: &mut [u32]) { out[..N].copy_from_slice(&self.p_key); out[N] = self.cipher; }
    fn from_flat(v: &[u32]) -> Self { let mut a = [0u32; N]; a.copy_from_slice(&v[..N]); TLWERep { cipher: v[N], p_key: a } }
}

pub struct TFHE<const TLWE_N: usize, const TRLWE_N: usize> { ctx: *mut sys::rtfhe_ctx, params: sys::rtfhe_params }

impl<const TLWE_N: usize, const TRLWE_N: usize> TFHE<TLWE_N, TRLWE_N> {
    /// `TFHE::new(s_key_tlwelv0, s_key_tlwelv1)` (tfhe.rs:21-25): generates KSK and BK for the given secret keys
    /// (host side; masks and noise from the OS CSPRNG, as the reference draws from thread_rng) and loads them on GPU 0.
    pub fn new(s_key_tlwelv0: [Binary; TLWE_N], s_key_tlwelv1: [Binary; TRLWE_N]) -> Self {
        Self::with_device(s_key_tlwelv0, s_key_tlwelv1, 0, None)
    }
    /// `key_seed = Some(seed)` is TEST ONLY: reproducible, not secure (rtfhe_keygen_with_keys_deterministic).
    pub fn with_device(s0: [Binary; TLWE_N], s1: [Binary; TRLWE_N], device: i32, key_seed: Option<u64>) -> Self {
        let mut p = sys::rtfhe_params { n: 0, N: 0, nbit: 0, l: 0, bgbit: 0, ks_t: 0, ks_basebit: 0 };
        unsafe { sys::rtfhe_default_params(&mut p) };
        p.n = TLWE_N as i32; p.N = TRLWE_N as i32; p.nbit = (TRLWE_N as u32).trailing_zeros() as i32;
        let k0: Vec<i32> = s0.iter().map(|&b| b as i32).collect();
        let k1: Vec<i32> = s1.iter().map(|&b| b as i32).collect();
        let mut bk = vec![0u32; TLWE_N * 2 * 2 * p.l as usize * TRLWE_N];
        let mut ksk = vec![0u32; TRLWE_N * p.ks_t as usize * ((1usize << p.ks_basebit) - 1) * (TLWE_N + 1)];
        let mut ctx: *mut sys::rtfhe_ctx = std::ptr::null_mut();
        unsafe {
            Self::check(std::ptr::null(), match key_seed {
                None => sys::rtfhe_keygen_with_keys(&p, k0.as_ptr(), k1.as_ptr(), bk.as_mut_ptr(), ksk.as_mut_ptr()),
                Some(seed) => sys::rtfhe_keygen_with_keys_deterministic(&p, seed, k0.as_ptr(), k1.as_ptr(), bk.as_mut_ptr(), ksk.as_mut_ptr()),
            });
            Self::check(std::ptr::null(), sys::rtfhe_ctx_create(&p, device as c_int, &mut ctx));
            Self::check(ctx, sys::rtfhe_load_bk_torus(ctx, bk.as_ptr()));
            // in through the reference's own container shape [[TLWERep; IKS_T = 4]; IKS_L = 8] (tlwe.rs:243-245), entry t = 4 included
            let mut ksk_ref = vec![0u32; TRLWE_N * p.ks_t as usize * (1usize << p.ks_basebit) * (TLWE_N + 1)];
            Self::check(std::ptr::null(), match key_seed {
                None => sys::rtfhe_ksk_expand_ref(&p, k0.as_ptr(), k1.as_ptr(), ksk.as_ptr(), ksk_ref.as_mut_ptr()),
                Some(seed) => sys::rtfhe_ksk_expand_ref_deterministic(&p, seed ^ 0x4b534b, k0.as_ptr(), k1.as_ptr(), ksk.as_ptr(), ksk_ref.as_mut_ptr()),
            });
            Self::check(ctx, sys::rtfhe_load_ksk_ref(ctx, ksk_ref.as_ptr()));
        }
        TFHE { ctx, params: p }
    }
    /// For keys the reference crate has ALREADY generated: `bk_f` = `BootstrappingKey(Vec<TRGSWRepF>)` flattened to
    /// `f64[n][2][6][N]` (per TRGSW `cipher_f[0..6]` then `pkey_f[0..6]`, tfhe.rs:116, trgsw.rs:64-67) and `ksk` = the inner Vec of
    /// `KeySwitchingKey(Vec<[[TLWERep<M>; IKS_T]; IKS_L]>)` exactly as it is, 4 entries per level (tlwe.rs:178-180, 243-245).
    /// Every `TLWERep` is written a[0..n) then b (its Rust field order is not `repr(C)`), level by level, entry by entry.
    pub fn from_reference_keys(bk_f: &[f64], ksk: &[[[TLWERep<TLWE_N>; 4]; 8]], device: i32) -> Self {
        let mut p = sys::rtfhe_params { n: 0, N: 0, nbit: 0, l: 0, bgbit: 0, ks_t: 0, ks_basebit: 0 };
        unsafe { sys::rtfhe_default_params(&mut p) };
        p.n = TLWE_N as i32; p.N = TRLWE_N as i32; p.nbit = (TRLWE_N as u32).trailing_zeros() as i32;
        assert_eq!(ksk.len(), TRLWE_N);
        assert_eq!(bk_f.len(), TLWE_N * 2 * 2 * p.l as usize * TRLWE_N);
        let w = TLWE_N + 1;
        let mut flat = vec![0u32; TRLWE_N * 8 * 4 * w];
        for (i, ks_i) in ksk.iter().enumerate() {
            for (l, ks_i_l) in ks_i.iter().enumerate() {
                for (t, rep) in ks_i_l.iter().enumerate() {
                    let at = ((i * 8 + l) * 4 + t) * w;
                    rep.write_flat(&mut flat[at..at + w]);
                }
            }
        }
        let mut ctx: *mut sys::rtfhe_ctx = std::ptr::null_mut();
        unsafe {
            Self::check(std::ptr::null(), sys::rtfhe_ctx_create(&p, device as c_int, &mut ctx));
            Self::check(ctx, sys::rtfhe_load_bk_fft(ctx, bk_f.as_ptr()));
            Self::check(ctx, sys::rtfhe_load_ksk_ref(ctx, flat.as_ptr()));
        }
        TFHE { ctx, params: p }
    }
    fn check(ctx: *const sys::rtfhe_ctx, rc: c_int) {
        if rc != 0 {
            let msg = unsafe { CStr::from_ptr(sys::rtfhe_last_error(ctx)) }.to_string_lossy().into_owned();
            panic!("rtfhe: {} (code {})", msg, rc);   // the reference has no Result on this path: failure = panic
        }
    }
    fn gate(&self, op: c_int, in0: &[TLWERep<TLWE_N>], in1: Option<&[TLWERep<TLWE_N>]>) -> Vec<TLWERep<TLWE_N>> {
        let w = TLWE_N + 1;
        let flat = |v: &[TLWERep<TLWE_N>]| { let mut f = vec![0u32; v.len() * w]; for (g, t) in v.iter().enumerate() { t.write_flat(&mut f[g * w..(g + 1) * w]); } f };
        let f0 = flat(in0);
        let f1 = in1.map(flat);
        let mut out = vec![0u32; in0.len() * w];
        let rc = unsafe { sys::rtfhe_gate_batch(self.ctx, op, f0.as_ptr(), f1.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()), out.as_mut_ptr(), in0.len()) };
        Self::check(self.ctx, rc);
        out.chunks(w).map(TLWERep::from_flat).collect()
    }
    pub fn hom_nand(&self, a: TLWERep<TLWE_N>, b: TLWERep<TLWE_N>) -> TLWERep<TLWE_N> { self.gate(sys::RTFHE_NAND, &[a], Some(&[b])).pop().unwrap() } // tfhe.rs:41-47
    pub fn hom_and(&self, a: TLWERep<TLWE_N>, b: TLWERep<TLWE_N>) -> TLWERep<TLWE_N> { self.gate(sys::RTFHE_AND, &[a], Some(&[b])).pop().unwrap() }   // tfhe.rs:48-54
    pub fn hom_or(&self, a: TLWERep<TLWE_N>, b: TLWERep<TLWE_N>) -> TLWERep<TLWE_N> { self.gate(sys::RTFHE_OR, &[a], Some(&[b])).pop().unwrap() }     // tfhe.rs:55-61
    pub fn hom_xor(&self, a: TLWERep<TLWE_N>, b: TLWERep<TLWE_N>) -> TLWERep<TLWE_N> { self.gate(sys::RTFHE_XOR, &[a], Some(&[b])).pop().unwrap() }   // tfhe.rs:62-68
    pub fn hom_not(&self, a: TLWERep<TLWE_N>) -> TLWERep<TLWE_N> { self.gate(sys::RTFHE_NOT, &[a], None).pop().unwrap() }                               // tfhe.rs:69-71
    /// (input_1 & control) | (input_0 & !control), tfhe.rs:27-40
    pub fn hom_mux(&self, control: TLWERep<TLWE_N>, input_0: TLWERep<TLWE_N>, input_1: TLWERep<TLWE_N>) -> TLWERep<TLWE_N> {
        let w = TLWE_N + 1;
        let (mut c, mut i0, mut i1, mut o) = (vec![0u32; w], vec![0u32; w], vec![0u32; w], vec![0u32; w]);
        control.write_flat(&mut c); input_0.write_flat(&mut i0); input_1.write_flat(&mut i1);
        Self::check(self.ctx, unsafe { sys::rtfhe_mux_batch(self.ctx, c.as_ptr(), i0.as_ptr(), i1.as_ptr(), o.as_mut_ptr(), 1) });
        TLWERep::from_flat(&o)
    }
    /// the reason for the engine: `count` independent gates, one kernel launch
    pub fn hom_nand_batch(&self, a: &[TLWERep<TLWE_N>], b: &[TLWERep<TLWE_N>]) -> Vec<TLWERep<TLWE_N>> {
        assert_eq!(a.len(), b.len());
        self.gate(sys::RTFHE_NAND, a, Some(b))
    }
    pub fn params(&self) -> &sys::rtfhe_params { &self.params }
}

impl<const TLWE_N: usize, const TRLWE_N: usize> Drop for TFHE<TLWE_N, TRLWE_N> {
    fn drop(&mut self) { unsafe { sys::rtfhe_ctx_destroy(self.ctx) } }
}
