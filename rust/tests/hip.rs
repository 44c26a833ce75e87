// Integration tests a maintainer adds next to tests/corpora.rs: the reference's own assertions
// (round trip + returned counts, tests/corpora.rs:32-85; doc-test src/lib.rs:23-39) against the
// `hip` module, plus equality with the pure-Rust coder -- the check this repository cannot run
// itself (no rustc in its image) and that closes its "parity unpinned" gap.
extern crate redux;

use std::io::Cursor;
use redux::model::{AdaptiveTreeModel, Parameters};

fn both(data: &[u8], p: &Parameters) {
    let mut cpu = Vec::new();
    let (ci, co) = redux::compress(&mut Cursor::new(data), &mut cpu, AdaptiveTreeModel::new(p.clone())).unwrap();
    let mut gpu = Vec::new();
    let (gi, go) = redux::hip::compress(&mut Cursor::new(data), &mut gpu, p).unwrap();
    assert_eq!((ci, co), (gi, go));
    assert_eq!(cpu, gpu);
    let mut back = Vec::new();
    let (di, dn) = redux::hip::decompress(&mut Cursor::new(&gpu), &mut back, p).unwrap();
    assert_eq!((di, dn), (go, data.len() as u64));
    assert_eq!(&back[..], data);
}

#[test]
fn doc_test_roundtrip() {
    both(b"redux", &Parameters::new(8, 14, 16).unwrap());
}

#[test]
fn blocks_equal_per_block_compress() {
    let p = Parameters::new(8, 30, 32).unwrap();
    let data: Vec<u8> = (0..200_000u32).map(|i| (i.wrapping_mul(2654435761) >> 24) as u8).collect();
    let (out, offs) = redux::hip::compress_blocks(&data, 65536, &p).unwrap();
    for (b, chunk) in data.chunks(65536).enumerate() {
        let mut cpu = Vec::new();
        redux::compress(&mut Cursor::new(chunk), &mut cpu, AdaptiveTreeModel::new(p.clone())).unwrap();
        assert_eq!(&out[offs[b] as usize..offs[b + 1] as usize], &cpu[..]);
    }
    let (back, sizes) = redux::hip::decompress_blocks(&out, &offs, 65536, &p).unwrap();
    for (b, chunk) in data.chunks(65536).enumerate() {
        assert_eq!(&back[b * 65536..][..sizes[b] as usize], chunk);
    }
}

// tests/corpora.rs:32-85 codes file by file; the batch call codes all of them in one launch and must give, block for
// block, what compress_blocks gives for each input alone.
#[test]
fn batch_equals_per_input_calls() {
    let p = Parameters::new(8, 30, 32).unwrap();
    let a: Vec<u8> = (0..150_000u32).map(|i| (i.wrapping_mul(2654435761) >> 24) as u8).collect();
    let b: Vec<u8> = Vec::new();
    let c: Vec<u8> = (0..70_001u32).map(|i| (i % 7) as u8).collect();
    let inputs: Vec<&[u8]> = vec![&a, &b, &c];
    let (out, offs, first) = redux::hip::compress_blocks_v(&inputs, 65536, &p).unwrap();
    for (i, x) in inputs.iter().enumerate() {
        let (one, of1) = redux::hip::compress_blocks(x, 65536, &p).unwrap();
        for k in 0..of1.len() - 1 {
            let g = first[i] as usize + k;
            assert_eq!(&out[offs[g] as usize..offs[g + 1] as usize], &one[of1[k] as usize..of1[k + 1] as usize]);
        }
    }
    let lens: Vec<u64> = inputs.iter().map(|x| x.len() as u64).collect();
    let back = redux::hip::decompress_blocks_v(&out, &offs, &lens, 65536, &p).unwrap();
    assert_eq!(&back[..a.len()], &a[..]);
    assert_eq!(&back[a.len()..], &c[..]);
}
